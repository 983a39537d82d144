set -e
export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_packed.so
python -m pytest tests/test_gpu_gp.py tests/test_gpu_scasml.py -m gpu -q -x > gpurun_out/r02_gputest7.log 2>&1 || (tail -30 gpurun_out/r02_gputest7.log; exit 1)
tail -3 gpurun_out/r02_gputest7.log
for v in packed default packed default; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'], j['l2_rel_error']['solver_gpu'])"
done
