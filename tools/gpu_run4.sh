set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1
./tools/microbench/bin/stamp_strip | grep -v trailing
python bench.py --steps 2000 --warmup 200 --cpu-seconds 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || (tail -20 gpurun_out/bench.err; exit 1)
cat gpurun_out/bench.json
bash tools/gpu_prof.sh
