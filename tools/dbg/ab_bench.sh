# A/B of bench.py between the product library and variant builds of it, alternating inside ONE GPU session
# (box-to-box differences are larger than the effects): bash tools/dbg/ab_bench.sh name1 name2 ...  (tools/microbench/bin/<name>.so)
mkdir -p gpurun_out; OUT=gpurun_out/ab_bench.jsonl; : > $OUT
for rep in 1 2 3; do
  for v in product "$@"; do
    if [ "$v" = product ]; then unset QBM_HIP_LIB; else export QBM_HIP_LIB=$PWD/tools/microbench/bin/$v.so; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib':'$v','rep':$rep,'us_per_step':round(d['ms_per_step']*1e3,3),'kernels_us':r['all_kernels_us']}))" >> $OUT
  done
done
unset QBM_HIP_LIB
cat $OUT
