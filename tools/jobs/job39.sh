set -e
for v in default icdf default icdf; do
  if [ $v == default ]; then unset SCASML_HIP_LIB; else export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gp-train-large 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['kernel_ms'])" || echo "$v failed"
done
unset SCASML_HIP_LIB
python bench.py --solver mlp --d 20 --level 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mlp default', j['ms_per_step'], j['value'])"
export SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_icdf.so
python bench.py --solver mlp --d 20 --level 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mlp icdf', j['ms_per_step'], j['value'])"
