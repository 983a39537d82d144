set -e
python -m pytest tests/test_gpu_scasml.py tests/test_gpu_full_size.py tests/test_gpu_gp.py -m gpu -q -x 2>&1 | tail -3
for v in default nointer default nointer; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])"
done
unset SCASML_HIP_LIB
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large --variant fh --level 4 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4', j['ms_per_step'], j['kernel_ms'], j['value'])"
