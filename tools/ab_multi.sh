#!/bin/bash
# usage (GPU box): tools/ab_multi.sh <reps> -- library A (working tree) against B (tools/build_B.sh) on the MobileNets, both modes
reps=${1:-2}
for m in mobilenetv3_large mobilenetv2 mobilenetv3_small; do
  bash tools/ab_lib.sh $reps python tools/quick_rate.py $m 1024 bf16 f32 || exit 1
done
