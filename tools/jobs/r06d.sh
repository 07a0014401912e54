cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-training --no-nonar --no-gblock --no-fast-leg --no-roofline --no-batch-sweep"
for rep in 1 2; do
for dm in 62 64; do
  for prec in bf16x3 f32; do
    HIFICAR_AR_DUAL_MAX=$dm $B --precision $prec 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dual_max=$dm', '$prec', d['value'], d['ms_per_step'])"
  done
done
done
