#!/bin/bash
# round-5 GPU batch: the whole -m gpu suite, the Cholesky A/B, the bench line (+ XL training legs), the distributed fit at M = 70 001 under RCCL
set -o pipefail
out=gpurun_out
python -u -m pytest tests -m gpu -x -q --durations=25 --deselect tests/test_gpu_xl.py > $out/r05_suite.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_suite.out; tail -5 $out/r05_suite.out
[ $rc -ne 0 ] && exit $rc
for M in 4224 35008; do
  python tools/chol_bench.py $M 5 >> $out/r05_chol_ab.txt 2>&1 || exit 1
  SCASML_HIP_LIB=$PWD/scasml_gp_amd/lib/libscasml_hip_splitdiag.so python tools/chol_bench.py $M 5 2>&1 | sed 's/^/[split diag + panel launches] /' >> $out/r05_chol_ab.txt || exit 1
done
cat $out/r05_chol_ab.txt
python bench.py --steps 20 --warmup 5 --gp-train-xl > $out/r05_bench_line_xl.json 2> $out/r05_bench_xl.err || { tail -20 $out/r05_bench_xl.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_bench_line_xl.json"))
print("value", j["value"], "ms", j["ms_per_step"], j["kernel_ms"])
for r in j["other_runs"]: print(r["workload"], r["ms_per_step"], r["value"])
for g in j["gp_train"]: print(g)
PY
python tools/dist_gp_demo.py --ranks 1 --backend nccl --n-dom 16667 --n-bdy 3333 > $out/r05_dist_gp_70k_rccl_one_rank.json 2> $out/r05_dist_gp_70k.err || { tail -20 $out/r05_dist_gp_70k.err; exit 1; }
cat $out/r05_dist_gp_70k_rccl_one_rank.json
python tools/dist_gp_demo.py --ranks 2 --n-dom 16667 --n-bdy 3333 --factor-only > $out/r05_dist_gp_70k_two_ranks_factor.json 2>> $out/r05_dist_gp_70k.err || { tail -20 $out/r05_dist_gp_70k.err; exit 1; }
cat $out/r05_dist_gp_70k_two_ranks_factor.json
