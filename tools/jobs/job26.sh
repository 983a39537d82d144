set -e
python -m pytest tests/test_gpu_scasml.py tests/test_gpu_configs.py tests/test_gpu_compat.py tests/test_gpu_equations.py tests/test_golden.py tests/test_gpu_bench_contract.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large --variant fh --level 4 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4', j['ms_per_step'], j['kernel_ms'], j['value'])"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('quad3', j['ms_per_step'], j['kernel_ms'])"
