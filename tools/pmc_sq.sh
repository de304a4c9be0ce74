# usage (on the GPU box): bash tools/pmc_sq.sh <model> <batch> <tag> "<counters>"
# one rocprofv3 --pmc pass (<= 8 SQ counters) over a short bench run; per-kernel CSV under gpurun_out/sq_<tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=$1; B=$2; T=$3; C=$4
mkdir -p $R/gpurun_out/sq_$T
timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq_$T -- python3 $R/bench.py --model $M --batch $B --steps 2 --warmup 1 --no-search --no-cpu-baseline --no-layer-profile > /dev/null 2>&1
echo done
