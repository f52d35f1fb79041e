# A/B of sample_Q on the conv-QBM QUBOs (tools/qubo_bench.py --conv) between the product library and variant builds of it,
# alternating inside ONE GPU session:  VARIANTS="product name1" bash tools/dbg/ab_csr.sh
mkdir -p gpurun_out; OUT=gpurun_out/ab_csr.jsonl; : > $OUT
for rep in 1 2; do
  for v in ${VARIANTS:-product}; do   # VARIANTS="product name1 name2": tools/microbench/bin/<name>.so (tools/dbg/build_variant.py)
    unset QBM_HIP_LIB
    [ $v = product ] || export QBM_HIP_LIB=$PWD/tools/microbench/bin/$v.so
    timeout -k 10 120 python tools/qubo_bench.py --conv --sweeps 1000 --reps 3 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(json.dumps({'lib':'$v','rep':$rep,'what':d['what'][20:40],'kernel':d['kernel'],'ms':d['ms_per_call']}))" >> $OUT
  done
done
cat $OUT
