set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/bench_configs.py 2>/dev/null | grep -i "disc\|sample_class_given"
