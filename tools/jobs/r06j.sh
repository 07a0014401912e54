cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_disc.py tests/test_gpu_disc_fuzz.py tests/test_gpu_gan.py tests/test_gpu_recipe.py -x -q -m gpu > gpurun_out/r06j_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r06j_tests.log | tail -2
for i in 1 2 3; do for lib in libhificar_base.so libhificar.so; do
HIFICAR_LIB=$PWD/articulatory_amd/$lib python tools/gan_bench.py --steps 10 2>/dev/null | grep "GAN iteration" | cut -c1-100 | sed "s/^/$lib /"
done; done
timeout 900 bash tools/pmc_by_layer.sh r06j gan > /dev/null 2>&1
grep "pack_all\|param_gather" gpurun_out/r06j_gan_pmc_hbm_by_layer.csv
