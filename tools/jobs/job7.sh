set -e
python -m pytest tests/test_gpu_scasml.py tests/test_gpu_full_size.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r02_gputest5.log 2>&1 || (tail -30 gpurun_out/r02_gputest5.log; exit 1)
tail -3 gpurun_out/r02_gputest5.log
for v in default prio1 prio2; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])"; done
done
unset SCASML_HIP_LIB
python -m pytest tests/test_gpu_dist_gp.py -m gpu -q -x -k "three_ranks or world1" > gpurun_out/r02_gputest6.log 2>&1 || (tail -30 gpurun_out/r02_gputest6.log; exit 1)
tail -3 gpurun_out/r02_gputest6.log
