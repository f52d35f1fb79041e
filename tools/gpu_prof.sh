set -e
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_cur
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cur -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --graph 0 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err || (tail -20 $OUT/prof.err; exit 1)
f=$(ls -t $(find $OUT/prof_cur -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):9.0f} min={r['MinNs']:>7s} max={r['MaxNs']:>7s} pct={r['Percentage']}")
PY
