#!/bin/bash
# usage: tools/chunk_ab.sh [model ...]  -- chunked expand+dw+1x1 kernel (dwc_kernel) on its measured shapes / off (FR_DWP_CHUNK=1 / 0), img/s at batch 1024
for m in ${@:-mobilenetv3_large mobilenetv2 mobilenetv3_small}; do
  for f in 1 0 1 0; do
    echo -n "FR_DWP_CHUNK=$f  "
    FR_DWP_CHUNK=$f timeout -k 10 120 python tools/quick_rate.py $m 1024 bf16 f32 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' || exit 1
    echo
  done
done
