cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06b_parity.log 2>&1; tail -3 gpurun_out/r06b_parity.log
AB_GAN=1 bash tools/ab.sh libhificar_base.so libhificar.so > gpurun_out/r06b_ab.txt 2>&1; cat gpurun_out/r06b_ab.txt
timeout 600 bash tools/pmc_by_layer.sh r06b f32
timeout 900 bash tools/pmc_by_layer.sh r06b gan
