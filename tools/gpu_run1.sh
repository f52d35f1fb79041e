set -e
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 2000 --warmup 200 > gpurun_out/bench_graph8.json 2> gpurun_out/bench_graph8.err || (tail -20 gpurun_out/bench_graph8.err; exit 1)
cat gpurun_out/bench_graph8.json
python bench.py --steps 2000 --warmup 200 --graph 0 --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err || (tail -20 gpurun_out/bench_eager.err; exit 1)
cat gpurun_out/bench_eager.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --graph 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err || (tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof.err; exit 1)
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*stats*" | head
