# usage (on the GPU box): bash tools/halo_probe.sh   -- parity, then the sphere nets with the shared-pixel conv kernel
set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q -k "golden or sphere or conv or batch" --timeout 120 --timeout-method thread 2>&1 | tail -3
for m in "sphere64 512" "sphere36 512" "sphere20 512"; do
  set -- $m
  echo "== $1 b$2"
  timeout -k 10 200 python -u bench.py --model $1 --batch $2 --no-cpu-baseline --no-search --no-layer-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
timeout -k 10 200 python -u bench.py --model sphere64 --batch 512 --no-cpu-baseline --no-search --layers 2>&1 | grep -E "layer[1-4]\.2\.block\.0"
