# A/B of any JSON-lines tool between the product library and variant builds, alternating inside ONE GPU session:
#   REPS=2 bash tools/dbg/ab_cmd.sh "python tools/qubo_bench.py --sweeps 100" name1 name2 ...
# (name = tools/microbench/bin/<name>.so from tools/dbg/build_variant.py, through QBM_HIP_LIB; "product" = the shipped library)
CMD="$1"; shift
mkdir -p gpurun_out; OUT=gpurun_out/ab_cmd.jsonl; : > $OUT
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    unset QBM_HIP_LIB
    [ $v = product ] || export QBM_HIP_LIB=$PWD/tools/microbench/bin/$v.so
    timeout -k 10 ${LIMIT:-150} $CMD 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); d={k:v for k,v in d.items() if not isinstance(v,(dict,list))}; d['lib']='$v'; d['rep']=$rep; print(json.dumps(d))" >> $OUT
  done
done
unset QBM_HIP_LIB
cat $OUT
