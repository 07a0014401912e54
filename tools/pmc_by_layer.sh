#!/bin/bash
# Per-launch-shape HBM counters (tools/pmc_by_layer.py) on the GPU box, through gpurun from the repo root:  tools/pmc_by_layer.sh <tag> [gan|f32]
# gan: the serial GAN iteration (HIFICAR_DISC_STREAMS=0, 3 warm-up + 1 timed iteration, the last one kept); f32: one bench step of the headline leg.
tag=${1:-r06}
what=${2:-gan}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcl_${tag}_${what}
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
if [ "$what" = gan ]; then
  cmd="python $root/tools/gan_bench.py --steps 1"
  export HIFICAR_DISC_STREAMS=0
  extra="--last-iterations 1 --markers-per-iteration 2 --groups-json $root/gpurun_out/${tag}_gan_traffic_by_group.json"
else
  cmd="python $root/bench.py --precision f32 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training --no-nonar --no-gblock"
  extra=""
fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -f $out/$ctr.log
  HIFICAR_LAUNCH_LOG=$out/$ctr.log rocprofv3 --pmc $ctr --kernel-trace -f csv -d $out/$ctr -- $cmd > $out/$ctr.stdout 2> $out/$ctr.stderr
done
cd $root
python tools/pmc_by_layer.py --fetch $out/FETCH_SIZE --fetch-log $out/FETCH_SIZE.log --write $out/WRITE_SIZE --write-log $out/WRITE_SIZE.log $extra \
  --command "${cmd//$root\//}" --out $root/gpurun_out/${tag}_${what}_pmc_hbm_by_layer.csv > $out/join.stdout 2> $out/join.stderr
tail -3 $out/join.stderr
