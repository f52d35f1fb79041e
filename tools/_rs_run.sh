cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -20 gpurun_out/parity.log; exit 1; }
tail -2 gpurun_out/parity.log
for v in "" "--variant=ahead2:-DQBM_DMA_AHEAD=2"; do
  echo "=== variant [$v]"; timeout -k 10 120 python tools/stamp_kernels.py $v 2>&1 | grep -v amdgpu.ids || exit 1
done > gpurun_out/rs_stamps.txt 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_new.json 2>gpurun_out/bench_new.err || exit 1
for i in 1 2 3; do timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/bench_driver_cfg.json 2>>gpurun_out/bench_driver_cfg.err; done
