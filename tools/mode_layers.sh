#!/bin/bash
# usage: tools/mode_layers.sh <tag> [ENV=VAL ...]   -- per-layer table of one precision mode (mobilenetv3_large batch 1024)
tag=$1; shift
prec=${PREC:-f24}
env "$@" timeout -k 10 300 python bench.py --precision $prec --layers --no-search --no-sphere64 --no-pairs --no-cpu-baseline --no-modes > gpurun_out/lay_$tag.json 2> gpurun_out/lay_$tag.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/lay_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernel_families'].items()})
PY
