ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | grep -i "WAIT\|IFETCH\|INST_LEVEL" | tr -s ' \t' ' ' | tr '\n' ';' ; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1); rm -rf $ROOT/gpurun_out/ipmc_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/ipmc_$tag -- python3 $ROOT/bench.py --steps 200 --warmup 20 --graph 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/ipmc_$tag.err
  echo "pmc $tag rc=$?"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ipmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for short in ("k_hidden", "k_visible", "k_grad"):
            if short in r["Kernel_Name"]:
                vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in vals.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()})
PY
