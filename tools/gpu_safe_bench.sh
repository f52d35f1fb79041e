mkdir -p gpurun_out
echo smoke > gpurun_out/progress.log
timeout -k 5 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/progress.log; tail -1 gpurun_out/smoke.log
timeout -k 5 200 python bench.py --steps 2000 --warmup 200 --cpu-seconds 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/progress.log
cut -c1-220 gpurun_out/bench.json
timeout -k 5 200 bash tools/gpu_prof.sh | head -4
cat gpurun_out/progress.log
