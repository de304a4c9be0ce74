# usage (on the GPU box): bash tools/r3_profiles.sh   -- round-3 set: default bench line, rocprofv3 kernel stats of the five
# workloads, HBM counter traffic of the three MobileNet modes.  Each step has its own time limit; stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo bench done
cd /tmp && export TMPDIR=/tmp
C="--steps 10 --warmup 3 --no-search --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64 --no-pairs"
for P in bf16 f32 f24; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mnv3_$P -- python3 $R/bench.py --precision $P $C > $O/prof_mnv3_$P.json 2>/dev/null
  echo prof $P done
done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s64 -- python3 $R/bench.py --model sphere64 --batch 512 $C > $O/prof_s64.json 2>/dev/null
echo prof s64 done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_search -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-layer-profile --no-modes --no-sphere64 --no-pairs > $O/prof_search.json 2>/dev/null
echo prof search done
cd $R
for P in bf16 f32 f24; do bash tools/pmc_traffic.sh mobilenetv3_large 1024 $P; done
bash tools/pmc_traffic.sh sphere64 512 bf16
for P in bf16 f32 f24; do
  python tools/hbm_traffic.py $(ls gpurun_out/tr_f_mobilenetv3_large_$P/*/*counter_collection.csv | tail -1) $(ls gpurun_out/tr_w_mobilenetv3_large_$P/*/*counter_collection.csv | tail -1) 14 $O/hbm_traffic_mobilenetv3_large_b1024_$P.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mobilenetv3_large batch 1024 $P, round 3" > $O/hbm_traffic_$P.txt
done
python tools/hbm_traffic.py $(ls gpurun_out/tr_f_sphere64_bf16/*/*counter_collection.csv | tail -1) $(ls gpurun_out/tr_w_sphere64_bf16/*/*counter_collection.csv | tail -1) 14 $O/hbm_traffic_sphere64_b512_bf16.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sphere64 batch 512 bf16, round 3" > $O/hbm_traffic_s64.txt
echo all done
