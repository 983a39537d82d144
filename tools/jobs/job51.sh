set -e -o pipefail
for i in 1 2; do python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1; done
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['kernel_ms'])"; done
