# GPU tests with a hard per-step time limit and progress on disk (a hang costs two minutes, not the call's limit)
mkdir -p gpurun_out
timeout -k 5 150 python -m pytest tests -m gpu -x -v > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
exit $rc
