# usage (on the GPU box): bash tools/pmc_traffic.sh <model> <batch>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=$1; B=$2
mkdir -p $R/gpurun_out/tr_f_$M $R/gpurun_out/tr_w_$M
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_f_$M -- python3 $R/bench.py --model $M --batch $B --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-layer-profile > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_w_$M -- python3 $R/bench.py --model $M --batch $B --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-layer-profile > /dev/null 2>&1
echo done
