#!/bin/bash
# A/B several builds of libhificar (same ABI) inside one gpurun call:  tools/ab.sh libA.so libB.so ...
for i in 1 2; do for lib in "$@"; do HIFICAR_LIB=$GRAFT_REPO_ROOT/articulatory_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-check --no-roofline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"; done; done
