cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
HIFICAR_FUZZ_CASES=256 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_train_fuzz.py tests/test_gpu_disc_fuzz.py tests/test_gpu_gblock.py -q -m gpu > gpurun_out/r06f_fuzz.log 2>&1; tail -3 gpurun_out/r06f_fuzz.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06f_gpu_tests.log 2>&1; tail -2 gpurun_out/r06f_gpu_tests.log
