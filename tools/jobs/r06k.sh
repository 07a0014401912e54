cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06k_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r06k_gpu_tests.log | tail -3
