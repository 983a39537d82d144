set -e
for v in default accw5 accw6 accw8 default accw6; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])"
done
