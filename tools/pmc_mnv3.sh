# usage (on the GPU box): bash tools/pmc_mnv3.sh <precision>
# What binds the MobileNetV3 kernels: VALU issue, LDS, or memory waits?  Separate rocprofv3 --pmc passes (<= 8 SQ counters each)
# over a short mobilenetv3_large batch-1024 run; per-kernel means printed at the end.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-f32}
A="--model mobilenetv3_large --batch 1024 --precision $P --steps 2 --warmup 1 --no-search --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64"
pass() {
  mkdir -p $R/gpurun_out/sq_$1
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/sq_$1 -- python3 $R/bench.py $A > $R/gpurun_out/sq_$1.log 2>&1
  echo pass $1 done
}
pass mnv3_${P}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
pass mnv3_${P}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY"
cd $R && python3 - $P <<'PY'
import glob, csv, collections, sys
P = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for tag in ("a", "b"):
    for f in glob.glob(f"gpurun_out/sq_mnv3_{P}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0) * n[(kv[0], "SQ_BUSY_CYCLES")])[:14]:
    print(k, "launches", max(n[(k, c)] for c in d))
    print("   ", {c: round(v / n[(k, c)], 1) for c, v in sorted(d.items())})
PY
