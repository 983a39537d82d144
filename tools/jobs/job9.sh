set -e
python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r02_gputest8.log 2>&1 || (tail -40 gpurun_out/r02_gputest8.log; exit 1)
tail -15 gpurun_out/r02_gputest8.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms'], j['l2_rel_error'])"
