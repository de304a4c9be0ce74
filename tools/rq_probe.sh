# usage (on the GPU box): bash tools/rq_probe.sh   -- resident-query search kernel: parity tests, then its rate
set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q -k "search or topk or gallery or match or shard" --timeout 180 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python -u tools/search_probe.py 10000000 32768 3 2>&1 | tail -1
timeout -k 10 200 python -u tools/search_probe.py 10000000 65536 2 2>&1 | tail -1
timeout -k 10 200 python -u tools/search_probe.py 2000000 4096 3 2>&1 | tail -1
