# HBM traffic of the strip kernels from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass), kilobytes; on gfx950 FETCH_SIZE
# reports half of the bytes of 16-B-per-lane streaming reads (global_load and buffer_load ... lds alike): doubled.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --graph 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err || (tail -5 $OUT/pmc_$c.err; exit 1)
done
python3 - <<'PY'
import csv, glob, json, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'{out}/pmc_{c}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            for short in ('k_hidden', 'k_visible', 'k_grad'):
                if short in k:
                    res[short][c + '_KB_avg'] = sum(v) / len(v)
                    res[short]['launches'] = len(v)
for k, d in res.items():
    fetch = d.get('FETCH_SIZE_KB_avg', 0.0) * 1024 * 2      # gfx950: x2 for wide streaming reads
    write = d.get('WRITE_SIZE_KB_avg', 0.0) * 1024
    d['hbm_bytes_per_launch'] = fetch + write
    d['fetch_bytes_corrected'] = fetch
    d['write_bytes'] = write
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --graph 0, CD-1 784x512 B=256; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)',
           'kernels': res}, open(out + '/pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
