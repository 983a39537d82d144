set -e
python -m pytest tests/test_gpu_gp.py tests/test_gpu_scasml.py tests/test_gpu_full_size.py tests/test_gpu_configs.py tests/test_gpu_compat.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms'], j['l2_rel_error']['solver_gpu'])"; done
