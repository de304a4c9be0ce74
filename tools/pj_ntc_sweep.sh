for prec in f32 bf16; do
for cfg in "0 256" "3 256" "2 256" "3 1000" "2 1000"; do set -- $cfg
echo "== $prec FR_PJ_NTC=$1 SUPER=$2"
FR_PJ_NTC=$1 FR_PJ_NTC_SUPER=$2 python tools/layer_table.py mobilenetv3_large 512 $prec 2>/dev/null | awk '/pw_proj_kernel/{t+=$(NF-1); printf "%s:%s ", $1, $(NF-1)} /^sum/{print "\n  proj_sum", t, $0}'
done; done
