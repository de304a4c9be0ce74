# usage (on the GPU box): bash tools/pmc_search.sh  -- two separate counter passes over a short 1:N search (10 M rows x 32768 queries):
# SQ activity (profiles/r01_search_pmc_sq.txt) and HBM fetch bytes (profiles/r01_search_pmc_fetch.txt)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq_search $R/gpurun_out/tr_search
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/sq_search -- python3 $R/tools/search_probe.py 10000000 32768 1 > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_search -- python3 $R/tools/search_probe.py 10000000 32768 1 > /dev/null 2>&1
cd $R && python3 - <<'PY'
import glob, csv, collections
for d in ("sq_search", "tr_search"):
    f = sorted(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "search" in k:
            print(k, {c: round(x / n[(k, c)], 1) for c, x in v.items()}, "launches", max(n[(k, c)] for c in v))
            if "FETCH_SIZE" in v:
                print("  HBM fetch per launch (FETCH_SIZE KiB x 2, gfx950 correction): %.1f GB" % (v["FETCH_SIZE"] / n[(k, "FETCH_SIZE")] * 1024 * 2 / 1e9))
PY
