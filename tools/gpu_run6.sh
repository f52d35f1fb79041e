set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1
for f in tools/microbench/bin/stamp_*; do echo $f; $f | grep "==\|compute loop\|dma\|lifetime\|epilogue\|reduce"; done
python bench.py --steps 2000 --warmup 200 --cpu-seconds 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || (tail -20 gpurun_out/bench.err; exit 1)
cut -c1-200 gpurun_out/bench.json
bash tools/gpu_prof.sh | head -4
