#!/bin/bash
# usage: tools/stem_fuse_ab.sh  -- stem+dw+1x1 fusion on / off (FR_FUSE_STEM), img/s of the three MobileNets that have the pattern
for m in mobilenetv3_large mobilenetv2 mobilenetv1; do
  for f in 1 0; do
    echo "FR_FUSE_STEM=$f"
    FR_FUSE_STEM=$f timeout -k 10 120 python tools/quick_rate.py $m 1024 bf16 f32 || exit 1
  done
done
