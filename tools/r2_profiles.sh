# usage (on the GPU box): bash tools/r2_profiles.sh  -- round-2 default bench line + rocprofv3 kernel-trace stats of each workload
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
P="--no-search --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mnv3_bf16 -- python3 $R/bench.py --steps 10 --warmup 3 $P > $O/prof_mnv3_bf16.json 2>/dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mnv3_f32 -- python3 $R/bench.py --precision f32 --steps 10 --warmup 3 $P > $O/prof_mnv3_f32.json 2>/dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s64 -- python3 $R/bench.py --model sphere64 --batch 512 --steps 10 --warmup 3 $P > $O/prof_s64.json 2>/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_search -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64 > $O/prof_search.json 2>/dev/null
echo done
