set -e
python -m pytest tests/test_gpu_rng.py tests/test_golden.py tests/test_gpu_mlp.py tests/test_gpu_scasml.py -m gpu -q -x 2>&1 | tail -3
bash tools/jobs/job12.sh
