set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_disc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_disc -o disc -- python3 tools/prof_disc.py ${1:-10} > gpurun_out/prof_disc.log 2>&1
f=$(find gpurun_out/prof_disc -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):9.0f} min={r['MinNs']:>7s} max={r['MaxNs']:>7s} pct={r['Percentage']}")
PY
