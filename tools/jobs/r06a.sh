cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06a_gpu_tests.log 2>&1; tail -3 gpurun_out/r06a_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; tail -c 600 gpurun_out/r06a_bench.json
timeout 900 bash tools/pmc_by_layer.sh r06a gan
timeout 600 bash tools/pmc_by_layer.sh r06a f32
