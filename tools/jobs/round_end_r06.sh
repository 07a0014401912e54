cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
bash tools/run_round_end.sh r06 2>&1 | tail -3
grep -E "passed|failed" gpurun_out/r06_gpu_tests.log | tail -2
timeout 900 bash tools/pmc_by_layer.sh r06 gan
timeout 600 bash tools/pmc_by_layer.sh r06 f32
for i in 1 2 3; do python tools/gan_bench.py --steps 10 2>/dev/null | grep "GAN iteration" | cut -c1-100; done
