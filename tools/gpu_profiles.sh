# Evidence for profiles/: rocprofv3 kernel stats (bench + every other hot-path kernel), PMC passes (HBM traffic; MFMA busy),
# the bench line, the in-kernel stamps.  Output under gpurun_out/profiles/ (copy into profiles/ afterwards).
#   ROUND=r04 bash tools/gpu_profiles.sh
R=${ROUND:-r04}
ROOT=$GRAFT_REPO_ROOT; [ -z "$ROOT" ] && ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles; mkdir -p $OUT; P=$ROOT/gpurun_out/progress.log; : > $P
cd /tmp && export TMPDIR=/tmp
stats() {  # $1 tag, rest: program
  tag=$1; shift; rm -rf $ROOT/gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- "$@" > $ROOT/gpurun_out/prof_$tag.out 2> $ROOT/gpurun_out/prof_$tag.err
  echo "stats $tag rc=$?" >> $P
  f=$(ls -t $(find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv") 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${R}_${tag}_kernel_stats.csv
}
stats bench python3 $ROOT/bench.py --steps 800 --warmup 80 --no-cpu-baseline      # the bench as it is run: hipGraph replays
cp $ROOT/gpurun_out/prof_bench.out $OUT/${R}_bench_under_rocprof.json
stats other_configs python3 $ROOT/tools/bench_configs.py
cp $ROOT/gpurun_out/prof_other_configs.out $OUT/${R}_other_configs_under_rocprof.jsonl
stats train_loop python3 $ROOT/tools/bench_train_loop.py --reps 1 --no-dp      # one epoch each of ClassificationRBM.train_rbm (3 rules) / Disc_QBM.train_model
cp $ROOT/gpurun_out/prof_train_loop.out $OUT/${R}_train_loop_under_rocprof.jsonl
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1); rm -rf $ROOT/gpurun_out/pmc_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc_$tag -- python3 $ROOT/bench.py --steps 200 --warmup 20 --graph 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/pmc_$tag.err
  echo "pmc $tag rc=$?" >> $P
done
# the QUBO Gibbs kernel (96 % of a Disc_QBM iteration): instruction mix, LDS activity and bank conflicts
ROUND=$R bash $ROOT/tools/qubo_pmc.sh > $ROOT/gpurun_out/qubo_pmc.log 2>&1; echo "qubo pmc rc=$?" >> $P
cp $ROOT/gpurun_out/qubo_pmc.json $OUT/${R}_qubo_pmc.json
cd $ROOT
python3 - <<PY
import csv, glob, json, os, collections
root = "$ROOT"; out = "$OUT"; R = "$R"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(root + "/gpurun_out/pmc_*"):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for short in ("k_hidden", "k_visible", "k_grad", "k_hgrad"):
                if short in r["Kernel_Name"]:
                    vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic, mfma = {}, {}
for k, c in vals.items():
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    fetch = avg.get("FETCH_SIZE", 0.0) * 1024 * 2      # KB; gfx950 tallies 128-B requests of wide streaming reads at 64 B: doubled
    write = avg.get("WRITE_SIZE", 0.0) * 1024
    traffic[k] = {"FETCH_SIZE_KB_avg": avg.get("FETCH_SIZE"), "WRITE_SIZE_KB_avg": avg.get("WRITE_SIZE"), "fetch_bytes_corrected": fetch,
                  "write_bytes": write, "hbm_bytes_per_launch": fetch + write, "launches": len(c.get("FETCH_SIZE", []))}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        simd_cycles = 1024 * avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0     # 256 CUs x 4 SIMDs x (active cycles of one XCD)
        mfma[k] = {n: avg[n] for n in avg if n.startswith("SQ_") or n.startswith("GRBM")}
        mfma[k]["mfma_busy_over_simd_cycles"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else None
        mfma[k]["mfma_busy_over_wave_cycles"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * avg["SQ_WAVE_CYCLES"]) if avg.get("SQ_WAVE_CYCLES") else None
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around bench.py --graph 0 (CD-1 784x512, B=256); FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "kernels": traffic}, open(f"{out}/{R}_pmc_traffic.json", "w"), indent=1)
json.dump({"source": "rocprofv3 --pmc SQ_* and GRBM_GUI_ACTIVE (separate passes) around bench.py --graph 0; SQ_VALU_MFMA_BUSY_CYCLES counts cycles "
                     "(32 per v_mfma_f32_16x16x4_f32), SQ_WAVE_CYCLES counts quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs; kernels run slower under "
                     "counter collection than in the bench (clock, serialised dispatch): use the ratios",
           "kernels": mfma}, open(f"{out}/{R}_pmc_mfma.json", "w"), indent=1)
print(json.dumps(mfma, indent=1)); print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}))
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > /dev/null 2> gpurun_out/kt.err; python tools/step_timeline.py gpurun_out/kt > $OUT/${R}_step_timeline.json; echo "timeline rc=$?" >> $P; rm -rf gpurun_out/kt
timeout -k 10 100 python tools/stamp_kernels.py > $OUT/${R}_stamps.txt 2>&1; echo "stamps rc=$?" >> $P
timeout -k 10 300 python bench.py > $OUT/${R}_bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> $P
timeout -k 10 150 python bench.py --force-dp --no-cpu-baseline > $OUT/${R}_bench_forced_dp.json 2>> gpurun_out/bench.err; echo "bench dp rc=$?" >> $P
timeout -k 10 200 python tools/bench_configs.py > $OUT/${R}_other_configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?" >> $P
timeout -k 10 200 python tools/qbm_pipeline.py --time > $OUT/${R}_qbm_pipeline.jsonl 2> gpurun_out/qbm_pipeline.err; echo "qbm rc=$?" >> $P
timeout -k 10 120 python tools/qubo_bench.py > $OUT/${R}_qubo_bench.jsonl 2> gpurun_out/qubo_bench.err; echo "qubo bench rc=$?" >> $P
timeout -k 10 120 python tools/qubo_bench.py --conv > $OUT/${R}_qubo_conv.jsonl 2>> gpurun_out/qubo_bench.err; timeout -k 10 120 python tools/qubo_bench.py --conv --dense >> $OUT/${R}_qubo_conv.jsonl 2>> gpurun_out/qubo_bench.err; echo "qubo conv rc=$?" >> $P
timeout -k 10 300 python tools/bench_train_loop.py 2> gpurun_out/train_loop.err | grep "^{" > $OUT/${R}_train_loop.jsonl; echo "train loop rc=$?" >> $P
timeout -k 10 100 python tools/disc_step_time.py > $OUT/${R}_disc_step.jsonl 2> gpurun_out/disc_step.err; echo "disc step rc=$?" >> $P
cat $P; head -c 600 $OUT/${R}_bench.json
