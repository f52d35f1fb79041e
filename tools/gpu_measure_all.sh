# tests, smoke, bench (graph + eager), rocprof kernel stats, PMC traffic -- every step under its own hard limit
mkdir -p gpurun_out; P=gpurun_out/progress.log; : > $P
step() { echo "start $1" >> $P; shift; "$@"; echo "rc=$?" >> $P; }
step tests   timeout -k 5 200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
step smoke   timeout -k 5 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
step pmc     timeout -k 5 400 bash tools/gpu_pmc_traffic.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
step prof    timeout -k 5 300 bash tools/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -5 gpurun_out/prof.log
step bench   timeout -k 5 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
step eager   timeout -k 5 300 python bench.py --graph 0 --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; cut -c1-200 gpurun_out/bench_eager.json
step configs timeout -k 5 300 python tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; cat gpurun_out/configs.jsonl
cat $P
