#!/bin/bash
# usage: tools/chunk_tile_sweep.sh <model> <C> <cin> <layer index> <prec> tile...   -- time of one chunked expand+dw+1x1 launch per tile (512
# images).  Prints the sum of the three layers (expand, depthwise, 1x1) and "3" when all three ran in one launch: a tile that does not
# fit the LDS cap makes the plan fall back to the two-layer fusion, and the expand conv then shows up on its own line.
m=$1; C=$2; cin=$3; li=$4; prec=$5; shift 5
sum3() { awk -v li=$li '$1>=li-2 && $1<=li { t+=$(NF-1); if ($0 ~ /fused into/) f++ } END { printf "%.4f ms  fused_layers=%d\n", t, f+1 }'; }
echo "== $m layer $li ($cin -> $C) $prec"
echo -n "dwp_kernel (FR_DWP_CHUNK=0): "; FR_DWP_CHUNK=0 timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sum3
for t in "$@"; do
  echo -n "chunked tile $t: "
  FR_DWP_CHUNK=2 FR_DWP_CAP_KB=${CAP:-0} FR_DWP_TILE="$C,$cin,$t" timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sum3 || exit 1
done
