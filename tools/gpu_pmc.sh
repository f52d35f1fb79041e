set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -c . $OUT/counters_list.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/pmc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --graph 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$tag.err || (tail -5 $OUT/pmc_$tag.err; echo "pmc set failed: $set")
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
for d in sorted(glob.glob(out+'/pmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            if 'k_hidden' in k or 'k_grad' in k or 'k_visible' in k:
                print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
