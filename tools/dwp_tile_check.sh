#!/bin/bash
# usage: tools/dwp_tile_check.sh <model> <C> <cin> <layer index> <prec> tile...  -- like chunk_tile_sweep.sh for dwp_kernel (no chunking): sum of the
# fused layers' times per forced tile and how many layers ran in one launch (a tile that does not fit silently falls back to fewer)
m=$1; C=$2; cin=$3; li=$4; prec=$5; shift 5
sum3() { awk -v li=$li '$1>=li-2 && $1<=li { t+=$(NF-1); if ($0 ~ /fused into/) f++ } END { printf "%.4f ms  fused_layers=%d\n", t, f+1 }'; }
echo "== $m layer $li ($cin -> $C) $prec, dwp_kernel"
echo -n "table / cost model: "; FR_DWP_CHUNK=0 timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sum3
for t in "$@"; do
  echo -n "tile $t: "
  FR_DWP_CHUNK=0 FR_DWP_TILE="$C,$cin,$t" timeout -k 10 100 python tools/layer_table.py $m 512 $prec 2>/dev/null | sum3 || exit 1
done
