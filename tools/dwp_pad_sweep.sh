#!/bin/bash
# per-layer times of the fused (expand) depthwise layers for FR_DWP_PAD = 0..3
for prec in f32 bf16; do for pad in 0 1 2 3; do
FR_DWP_PAD=$pad timeout -k 10 300 python bench.py --precision $prec --layers --no-search --no-sphere64 --no-pairs --no-cpu-baseline --no-modes > gpurun_out/pad_${prec}_$pad.json 2> gpurun_out/pad_${prec}_$pad.txt || exit 1
echo "$prec pad $pad: $(grep dwp_kernel gpurun_out/pad_${prec}_$pad.txt | awk '{printf "%s ", $3}')"
done; done
