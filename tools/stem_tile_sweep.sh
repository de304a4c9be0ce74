#!/bin/bash
# usage: tools/stem_tile_sweep.sh <model> <C>  -- time of the stem+dw+1x1 launch per tile (FR_DWP_TILE), 512 images, bf16 and f32
m=${1:-mobilenetv3_large}; C=${2:-16}
for prec in bf16 f32; do
  echo "== $m $prec: unfused stem + dwp"
  FR_FUSE_STEM=0 timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sed -n '1,3p'
  for t in "4,28" "8,16" "8,28" "8,56" "14,16" "14,28" "16,16" "16,28" "16,32" "7,28" "7,56" "4,56" "2,56" "14,8" "16,8"; do
    echo "-- tile $t"
    FR_DWP_CAP_KB=150 FR_DWP_TILE="$C,3,$t" timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sed -n '3p' || exit 1
  done
done
