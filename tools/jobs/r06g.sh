cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mrf_mean" 2>&1 | tail -5
bash tools/run_round_end.sh r06 2>&1 | tail -5
timeout 900 bash tools/pmc_by_layer.sh r06 gan
timeout 600 bash tools/pmc_by_layer.sh r06 f32
