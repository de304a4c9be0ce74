# usage (on the GPU box): bash tools/pmc_traffic.sh <model> <batch> [precision]
# two separate counter passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only -- then tools/hbm_traffic.py with
# <steps+warmup> = 14: 3 timed + 1 warm-up + the 10 pre-warm steps bench.py runs when the layer profile is off
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=$1; B=$2; P=${3:-bf16}
mkdir -p $R/gpurun_out/tr_f_${M}_$P $R/gpurun_out/tr_w_${M}_$P
A="--model $M --batch $B --precision $P --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64 --no-pairs"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_f_${M}_$P -- python3 $R/bench.py $A > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_w_${M}_$P -- python3 $R/bench.py $A > /dev/null 2>&1
echo done $M $P
