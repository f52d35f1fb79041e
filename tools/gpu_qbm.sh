set -e
python -m pytest tests -m gpu -x -q -k "qubo or gibbs or qbm" 2>&1 | tail -3
python tools/bench_qbm.py --cpu-seconds 2
