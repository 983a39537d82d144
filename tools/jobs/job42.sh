set -e -o pipefail
python -m pytest tests -m gpu -q -x > gpurun_out/r02_gputest_icdf.log 2>&1 || (grep -v amdgpu.ids gpurun_out/r02_gputest_icdf.log | tail -40 | cut -c1-250; exit 1)
tail -3 gpurun_out/r02_gputest_icdf.log
python bench.py --steps 10 --warmup 3 --no-gp-train-large --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['kernel_ms'])"
python bench.py --solver mlp --d 20 --level 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mlp d20 n2', j['ms_per_step'], j['value'])"
python bench.py --variant fh --level 4 --steps 5 --warmup 2 --no-cpu-baseline --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4', j['ms_per_step'], j['value'], j['kernel_ms'])"
