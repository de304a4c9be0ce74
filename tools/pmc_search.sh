# usage (on the GPU box): bash tools/pmc_search.sh  -- HBM fetch bytes of a short 1:N search (10 M rows x 32768 queries)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tr_search
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_search -- python3 $R/tools/search_probe.py 10000000 32768 1 > /dev/null 2>&1
cd $R && python3 - <<'PY'
import glob, csv, collections
f = sorted(glob.glob("gpurun_out/tr_search/**/*counter_collection.csv", recursive=True))[-1]
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in tot:
    if "search" in k: print(k, "launches", n[k], "HBM fetch per launch (FETCH_SIZE KiB x 2, gfx950 correction): %.1f GB" % (tot[k] / n[k] * 1024 * 2 / 1e9))
PY
