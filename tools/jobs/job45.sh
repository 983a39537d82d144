set -e
python tools/chol_bench.py 4224 5 2>/dev/null
python tools/chol_bench.py 3008 5 2>/dev/null
python tools/chol_bench.py 8160 3 2>/dev/null
python -m pytest tests/test_gpu_gp.py tests/test_gpu_configs.py tests/test_gpu_equations.py tests/test_gpu_compat.py -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); [print(g) for g in j['gp_train']]"
