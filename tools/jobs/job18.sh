set -e
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dist_gp.py > gpurun_out/r02_gputest11.log 2>&1 || (tail -40 gpurun_out/r02_gputest11.log; exit 1)
tail -3 gpurun_out/r02_gputest11.log
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms'], j['l2_rel_error']['solver_gpu'])"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large --variant fh --level 4 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4', j['ms_per_step'], j['kernel_ms'])"
