#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r02
# Passes (each its own run; PMC passes use --kernel-trace only, as the pool requires):
#   kernel stats (both arithmetics), FETCH_SIZE, WRITE_SIZE, MFMA-busy / SQ cycle counters.  Output: gpurun_out/prof_<tag>_*/
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
common="--no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training --no-nonar --no-gblock"
rocprofv3 -L > $out/prof_${tag}_counters_available.txt 2>&1
for prec in f32 bf16x3; do
  rocprofv3 --kernel-trace --stats -f csv -d $out/prof_${tag}_${prec}_stats -- python $root/bench.py --precision $prec --steps 3 --warmup 1 $common > $out/prof_${tag}_${prec}_bench_under_rocprof.json 2> $out/prof_${tag}_${prec}_stats.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -f csv -d $out/prof_${tag}_${prec}_$ctr -- python $root/bench.py --precision $prec --steps 1 --warmup 0 --no-roofline $common > /dev/null 2> $out/prof_${tag}_${prec}_$ctr.log
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -f csv -d $out/prof_${tag}_${prec}_mfma -- python $root/bench.py --precision $prec --steps 1 --warmup 0 --no-roofline $common > /dev/null 2> $out/prof_${tag}_${prec}_mfma.log
done
# the secondary legs of the bench line (round 5), each in its own process: BASELINE config 2 (non-AR, batch 8) and GBlockGenerator (batch 64)
for leg in nonar gblock; do
  rocprofv3 --kernel-trace --stats -f csv -d $out/prof_${tag}_${leg}_stats -- python $root/tools/leg_bench.py --leg $leg --steps 3 --warmup 1 > $out/prof_${tag}_${leg}_bench_under_rocprof.json 2> $out/prof_${tag}_${leg}_stats.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -f csv -d $out/prof_${tag}_${leg}_$ctr -- python $root/tools/leg_bench.py --leg $leg --steps 1 --warmup 0 > /dev/null 2> $out/prof_${tag}_${leg}_$ctr.log
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -f csv -d $out/prof_${tag}_${leg}_mfma -- python $root/tools/leg_bench.py --leg $leg --steps 1 --warmup 0 > /dev/null 2> $out/prof_${tag}_${leg}_mfma.log
done
# training: generator train step and the full GAN iteration
rocprofv3 --kernel-trace --stats -f csv -d $out/prof_${tag}_train_stats -- python $root/tools/train_bench.py --steps 5 > $out/prof_${tag}_train_bench.txt 2> $out/prof_${tag}_train_stats.log
rocprofv3 --kernel-trace --stats -f csv -d $out/prof_${tag}_gan_stats -- python $root/tools/gan_bench.py --steps 5 > $out/prof_${tag}_gan_bench.txt 2> $out/prof_${tag}_gan_stats.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -f csv -d $out/prof_${tag}_train_mfma -- python $root/tools/train_bench.py --steps 1 > /dev/null 2> $out/prof_${tag}_train_mfma.log
HIFICAR_DISC_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -f csv -d $out/prof_${tag}_gan_mfma -- python $root/tools/gan_bench.py --steps 1 > /dev/null 2> $out/prof_${tag}_gan_mfma.log
# HBM traffic of the training kernels (round 4): the GAN iteration with the discriminators on the caller's stream (no overlap: per-launch counters;
# the engine keeps the tile shapes of the overlapped run)
for ctr in FETCH_SIZE WRITE_SIZE; do
  HIFICAR_DISC_STREAMS=0 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $out/prof_${tag}_gan_$ctr -- python $root/tools/gan_bench.py --steps 1 > /dev/null 2> $out/prof_${tag}_gan_$ctr.log
done
ls $out | grep prof_${tag}
