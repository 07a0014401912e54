#!/bin/bash
# A/B several builds of libhificar (same ABI) inside one gpurun call, interleaved, two rounds:  tools/ab.sh libA.so libB.so ...
# prints headline value / batch 1 / batch 8 per build (bench.py, 10 steps) and the GAN iteration (tools/gan_bench.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do for lib in "$@"; do
  HIFICAR_LIB=$root/articulatory_amd/$lib python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-training --no-nonar --no-gblock --no-fast-leg --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'batch64', d['value'], 'ms', d['ms_per_step'], ' '.join(f\"b{b['batch']}={b['value']:.0f}\" for b in d.get('batch_sweep', [])[:2]))"
  [ -n "$AB_GAN" ] && HIFICAR_LIB=$root/articulatory_amd/$lib python $root/tools/gan_bench.py --steps 10 2>/dev/null | grep "GAN iteration" | cut -c1-110 | sed "s/^/$lib /"
done; done
