#!/bin/bash
# usage: tools/batch_sweep.sh [model]  -- img/s of model(x) at batch 64..1024 in bf16 / f32 (is a MALL-sized batch faster per image?)
m=${1:-mobilenetv3_large}
for b in 64 128 256 512 1024; do
  timeout -k 10 120 python tools/quick_rate.py $m $b bf16 f32 || exit 1
done
