#!/bin/bash
set -o pipefail
out=gpurun_out
python -u -m pytest tests/test_gpu_full_size.py tests/test_gpu_configs.py -x -q -s --durations=6 > $out/r05_parity.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_parity.out; tail -12 $out/r05_parity.out | cut -c1-300
[ $rc -ne 0 ] && exit $rc
grep "explained parity" $out/r05_parity.out > $out/r05_explained_parity.txt
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_line.json ) 2> $out/r05_bench.time || exit 1
tail -4 $out/r05_bench.time
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05_bench_line.json") if l.startswith("{")][0])
print(j["value"], j["ms_per_step"], j["kernel_ms"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["vector"] and j["roofline"]["vector"]["sum_model_ms"], j["roofline_path"]["frac"])
PY
