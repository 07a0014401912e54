# A/B of the register-blocked (NB = 2) wave tiles: HIFICAR_NB = 0 (off) / 1 (cost model) / 2 (forced wherever a shape fits)
HIFICAR_NB=2 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for prec in f32 bf16x3; do for i in 1 2; do for nb in 0 1; do
 HIFICAR_NB=$nb python bench.py --precision $prec --steps 3 --warmup 1 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$prec NB=$nb', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_us'])"
done; done; done
