cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 200 gpurun_out/r06_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
t=d['training']
print(d['value'], d['roofline']['frac'], d['fast_bf16x3']['value'], t['gan_iteration_ms'], t['serial_iteration_ms'], t['host_enqueue_ms'])
PY
for i in 1 2; do python tools/gan_bench.py --steps 10 2>/dev/null | grep "GAN iteration" | cut -c1-100; done
