# GPU tests + smoke + default bench, every step under its own hard limit, progress on disk
mkdir -p gpurun_out; P=gpurun_out/progress.log; : > $P
timeout -k 10 ${TEST_LIMIT:-420} python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc" >> $P
tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" >> $P; tail -1 gpurun_out/smoke.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench rc=$rc" >> $P
cat gpurun_out/bench.json
exit $rc
