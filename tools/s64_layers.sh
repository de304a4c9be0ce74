#!/bin/bash
# usage: tools/s64_layers.sh <tag> [ENV=VAL ...]   -- per-layer table of sphere64 batch 512 bf16, summed per stage
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --model sphere64 --batch 512 --precision bf16 --layers --no-search --no-sphere64 --no-pairs --no-cpu-baseline --no-modes > gpurun_out/lay_$tag.json 2> gpurun_out/lay_$tag.txt || exit 1
python - <<PY
import json, re, collections
d = json.loads(open('gpurun_out/lay_$tag.json').read().strip().splitlines()[-1])
st = collections.OrderedDict()
for ln in open('gpurun_out/lay_$tag.txt'):
    m = re.match(r'(\S+)\s+(\S+)\s+([0-9.]+) ms', ln)
    if not m: continue
    name, ms = m.group(1), float(m.group(3))
    key = name.split('.')[0] + ('.0' if re.match(r'layer\d\.0$', name) else '')
    st[key] = st.get(key, 0.0) + ms
print('$tag', d['value'], d['ms_per_step'], {k: round(v, 4) for k, v in st.items()})
PY
