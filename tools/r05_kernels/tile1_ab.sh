#!/bin/bash
# A/B of the tile shape of the single-layer launches (upsamplers) at batch 64: HIFICAR_TILE1="cin,MI,WM,WN" forces a shape (dev override in launch_conv)
#   gpurun -- bash tools/tile1_ab.sh [batch]
b=${1:-64}
python tools/layer_profile.py --batch $b --steps 10 2>/dev/null | grep -E "total|upsamples"
for cin in 512 256 128 64; do
  for t in 4,1,4 4,2,2 4,4,1 2,1,4 2,2,2 2,4,1 1,1,4 1,2,2 1,4,1; do
    echo -n "cin $cin tile $t: "
    HIFICAR_TILE1="$cin,$t" python tools/layer_profile.py --batch $b --steps 10 2>/dev/null | grep -E "cin_match|upsamples" | awk -v c=$cin 'BEGIN{m[512]="upsamples.0";m[256]="upsamples.1";m[128]="upsamples.2";m[64]="upsamples.3"} index($0,m[c]){print $1, $4, $5}'
  done
done
