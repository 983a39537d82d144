set -e
for v in default read2 default read2; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  echo -n "$v: "; python tools/chol_bench.py 35008 2 2>/dev/null
done
unset SCASML_HIP_LIB
python -m pytest tests/test_gpu_gp.py tests/test_gpu_dist_gp.py -m gpu -q -x 2>&1 | tail -3
