# usage (on the GPU box): bash tools/pmc_conv.sh
# Is conv_glds_kernel bound by the vector-memory path (TA/TCP) or by MFMA issue?  Two separate --pmc passes over a
# short sphere64 run of the UNMODIFIED library, within the per-block counter limits (2 TA, 4 TCP, 1 GRBM | 8 SQ).
set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_sq.sh sphere64 512 conv_ta "TA_BUSY_avr TA_BUSY_max TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
bash $R/tools/pmc_sq.sh sphere64 512 conv_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
cd $R && python3 - <<'PY'
import glob, csv, collections
for tag in ("conv_ta", "conv_sq"):
    files = glob.glob(f"gpurun_out/sq_{tag}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print("==", tag, len(files), "file(s)")
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in d.items()}, "launches", max(n[(k, c)] for c in d))
PY
