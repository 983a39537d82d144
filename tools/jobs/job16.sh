set -e
python -m pytest tests/test_gpu_gp.py tests/test_gpu_configs.py tests/test_gpu_dist_gp.py -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_c.json
python -c "
import json; j=json.load(open('gpurun_out/r02_bench_c.json'))
print(j['ms_per_step'], j['kernel_ms'])
for g in j['gp_train']: print(g)
print(j['roofline_path'])
"
