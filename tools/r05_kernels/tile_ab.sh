#!/bin/bash
# A/B of forced tile shapes for the 3-branch ResBlock launches (HIFICAR_TILE="cin,MI,WM,WN"):  tools/tile_ab.sh "256,2,1,4" "64,2,2,2" ...
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  run base
  for t in "$@"; do HIFICAR_TILE=$t run "$t"; done
done
