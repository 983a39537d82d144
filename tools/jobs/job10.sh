set -e
python -m pytest tests/test_gpu_gp.py tests/test_gpu_configs.py tests/test_gpu_compat.py tests/test_gpu_equations.py -m gpu -q -x > gpurun_out/r02_gputest9.log 2>&1 || (tail -40 gpurun_out/r02_gputest9.log; exit 1)
tail -3 gpurun_out/r02_gputest9.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r02_bench_b.json
python -c "
import json; j=json.load(open('gpurun_out/r02_bench_b.json'))
print(j['ms_per_step'], j['kernel_ms'])
for g in j['gp_train']: print(g)
print(j['roofline']['valu_issue'], j['roofline']['traffic_source'])
"
