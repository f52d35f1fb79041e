# A/B of bench.py between the product library and variants, alternating inside ONE GPU session (box-to-box differences are
# larger than the effects): bash tools/dbg/ab_bench.sh name1 env:VAR=value ...
#   name       a variant build of the library, tools/microbench/bin/<name>.so (tools/dbg/build_variant.py), through QBM_HIP_LIB
#   env:VAR=v  the product library with environment variable VAR set to v for the run
mkdir -p gpurun_out; OUT=gpurun_out/ab_bench.jsonl; : > $OUT
for rep in 1 2 3; do
  for v in product "$@"; do
    unset QBM_HIP_LIB; EV=""
    case "$v" in
      product) ;;
      env:*) EV="${v#env:}";;
      *) export QBM_HIP_LIB=$PWD/tools/microbench/bin/$v.so;;
    esac
    env $EV timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib':'$v','rep':$rep,'us_per_step':round(d['ms_per_step']*1e3,3),'kernels_us':r['all_kernels_us']}))" >> $OUT
  done
done
unset QBM_HIP_LIB
cat $OUT
