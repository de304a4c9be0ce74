# usage (here, after gpurun merged gpurun_out/r3): bash tools/r3_collect_profiles.sh <tag>  -> profiles/r03_*_<tag>*
set -e
T=${1:-v1}
O=gpurun_out/r3
cp $O/bench_default.json profiles/r03_bench_$T.json
for w in mnv3_bf16:embed_mnv3large_b1024_bf16 mnv3_f32:embed_mnv3large_b1024_f32 mnv3_f24:embed_mnv3large_b1024_f24 s64:embed_sphere64_b512_bf16 search:search_10M_x_65536_bf16; do
  d=${w%%:*}; n=${w##*:}
  ks=$(ls $O/prof_$d/*/*_kernel_stats.csv | head -1); kt=$(ls $O/prof_$d/*/*_kernel_trace.csv | head -1)
  cp $ks profiles/r03_${n}_${T}_kernel_stats.csv
  python tools/summarize_rocprof.py $kt profiles/r03_${n}_${T}_by_kernel.csv
done
for P in bf16 f32 f24; do cp $O/hbm_traffic_mobilenetv3_large_b1024_$P.json profiles/hbm_traffic_mobilenetv3_large_b1024_$P.json; done
cp $O/hbm_traffic_sphere64_b512_bf16.json profiles/hbm_traffic_sphere64_b512_bf16.json
ls profiles | grep r03_
