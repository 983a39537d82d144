set -e
python -m pytest tests/test_gpu_gp.py tests/test_gpu_configs.py tests/test_gpu_dist_gp.py -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); [print({k:g[k] for k in ('M','fit_s','cholesky_ms','inverse_ms','inverse_tflops')}) for g in j['gp_train']]"
