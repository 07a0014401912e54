cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-nonar --no-gblock > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; tail -c 300 gpurun_out/r06h_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06h_bench.json').read().strip().splitlines()[-1])
t=d['training']
print(t['gan_iteration_ms'], t['serial_iteration_ms'])
for r in t['by_group']: print(r)
print('--')
for r in t['dominant_kernel_family_by_group']: print(r)
print(t['dominant_kernel_on_generator_layers'])
print(t['roofline']['dominant_kernel'])
PY
