set -e
mkdir -p gpurun_out
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --graph 0 > gpurun_out/bench_eager.json 2> gpurun_out/e1.err || (tail gpurun_out/e1.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_eager.json')); print('eager single:', d['ms_per_step'], 'ms/step')"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --force-dp > gpurun_out/bench_dp1.json 2> gpurun_out/e2.err || (tail gpurun_out/e2.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_dp1.json')); print('forced dp, 1 rank:', d['ms_per_step'], 'ms/step', d['config']['parallelism'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/e3.err || (tail gpurun_out/e3.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_torchrun1.json')); print('torchrun 1 rank:', d['ms_per_step'], 'ms/step', d['n_gpus'])"
