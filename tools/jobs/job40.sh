set -e
python -m pytest tests/test_gpu_rng.py tests/test_golden.py tests/test_gpu_mlp.py tests/test_gpu_scasml.py -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['kernel_ms'], j['cpu_baseline'])"
