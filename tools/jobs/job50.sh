python tools/em_debug.py 2>&1 | grep -v amdgpu | tail -4
python -m pytest tests/test_gpu_gp.py tests/test_gpu_scasml.py tests/test_golden.py tests/test_gpu_full_size.py tests/test_gpu_equations.py -m gpu -q 2>&1 | tail -4 | cut -c1-250
for v in default noem default noem; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])" || echo "$v failed"
done
