set -e
python tools/chol_bench.py 35008 2 2>/dev/null
python tools/chol_bench.py 16384 2 2>/dev/null
python -m pytest tests/test_gpu_gp.py tests/test_gpu_dist_gp.py -m gpu -q -x 2>&1 | tail -3
