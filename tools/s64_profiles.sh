# usage (on the GPU box): bash tools/s64_profiles.sh  -- sphere64 batch-512: bench line with roofline, kernel-trace stats, HBM traffic passes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s64
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --model sphere64 --batch 512 --no-search > $O/bench_s64.json 2> $O/bench_s64.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s64 -- python3 $R/bench.py --model sphere64 --batch 512 --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-layer-profile > $O/prof_s64.json 2>/dev/null
bash $R/tools/pmc_traffic.sh sphere64 512
echo done
