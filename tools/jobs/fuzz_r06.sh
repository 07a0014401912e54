cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
HIFICAR_FUZZ_CASES=256 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_train_fuzz.py tests/test_gpu_disc_fuzz.py tests/test_gpu_gblock.py -q -m gpu > gpurun_out/r06_fuzz.log 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/r06_fuzz.log | tail -5
