cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "bf16x3 or fuzz" > gpurun_out/r06m_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r06m_tests.log | tail -2
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-training --no-nonar --no-gblock --no-fast-leg --no-roofline --precision bf16x3"
for i in 1 2 3; do for lib in libhificar_base.so libhificar.so; do
HIFICAR_LIB=$PWD/articulatory_amd/$lib $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], [round(b['value']) for b in d.get('batch_sweep',[])[:2]])"
done; done
