# usage (here, after gpurun merged gpurun_out/r2): bash tools/r2_collect_profiles.sh <tag>  -> profiles/r02_*_<tag>*
set -e
T=${1:-v1}
O=gpurun_out/r2
cp $O/bench_default.json profiles/r02_bench_$T.json
for w in mnv3_bf16:embed_mnv3large_b1024_bf16 mnv3_f32:embed_mnv3large_b1024_f32 s64:embed_sphere64_b512_bf16 search:search_10M_x_65536_bf16; do
  d=${w%%:*}; n=${w##*:}
  ks=$(ls $O/prof_$d/*/*_kernel_stats.csv | head -1); kt=$(ls $O/prof_$d/*/*_kernel_trace.csv | head -1)
  cp $ks profiles/r02_${n}_${T}_kernel_stats.csv
  python tools/summarize_rocprof.py $kt profiles/r02_${n}_${T}_by_kernel.csv
done
ls profiles | grep r02_
