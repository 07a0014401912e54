#!/bin/bash
# Power draw and shader clock of the GPU while the headline bench runs (rocm-smi sampled next to it):
#   gpurun -- bash tools/power_probe.sh  -> gpurun_out/power_probe.txt
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/power_probe.txt
mkdir -p $(dirname $out)
rocm-smi --showmaxpower --showclocks --showpower > $out 2>&1
echo "=== samples while python bench.py --steps 8 runs (idle first, then the warm-up, the timed steps, the event pass)" >> $out
( for i in $(seq 1 400); do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')" >> $out; sleep 0.1
  done ) &
smp=$!
sleep 1
python bench.py --steps 8 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training > ${out%.txt}_bench.json 2>/dev/null
kill $smp 2>/dev/null
tail -c 400 ${out%.txt}_bench.json
wc -l $out
