set -e
for v in default ahead1 ahead2 ahead5 default ahead1; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])"
done
unset SCASML_HIP_LIB
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large --variant fh --level 4 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4 default', j['ms_per_step'], j['kernel_ms'])"
export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_ahead1.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large --variant fh --level 4 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fh4 ahead1', j['ms_per_step'], j['kernel_ms'])"
unset SCASML_HIP_LIB
python -m pytest tests/test_gpu_scasml.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -2
