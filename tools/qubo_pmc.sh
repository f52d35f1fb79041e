# PMC pass over the QUBO Gibbs kernel (one counter set per pass; program directly after --)
ROOT=$GRAFT_REPO_ROOT; [ -z "$ROOT" ] && ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1); rm -rf $ROOT/gpurun_out/qpmc_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/qpmc_$tag -- python3 $ROOT/tools/qubo_bench.py --sweeps 100 --reps 2 > $ROOT/gpurun_out/qpmc_$tag.out 2> $ROOT/gpurun_out/qpmc_$tag.err
  echo "pmc $tag rc=$?"
done
cd $ROOT
python3 - <<PY
import csv, glob, json, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$ROOT/gpurun_out/qpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_qubo_gibbs" in r["Kernel_Name"]:
            key = "k_qubo_gibbs_lds<%s>" % r["Kernel_Name"].split("<")[1].split(">")[0] if "<" in r["Kernel_Name"] else r["Kernel_Name"]
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in vals.items()}
for k, c in out.items():
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_bank_conflict_over_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("GRBM_GUI_ACTIVE"):
        cu_cycles = 256 * c["GRBM_GUI_ACTIVE"] / 8.0
        c["lds_active_over_cu_cycles"] = c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles
        c["valu_insts_per_simd_cycle"] = c.get("SQ_INSTS_VALU", 0) / (4 * cu_cycles)
print(json.dumps(out, indent=1))
json.dump(out, open("$ROOT/gpurun_out/qubo_pmc.json", "w"), indent=1)
PY
