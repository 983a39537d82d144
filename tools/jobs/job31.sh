set -e
for v in default notail neither; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  echo -n "$v: "; python tools/chol_bench.py 35008 2 2>/dev/null
done
