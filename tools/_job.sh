#!/bin/bash
set -o pipefail
out=gpurun_out
python -u -m pytest tests -m gpu -x -q -k "shard or split or bench or rank or units or jax_stream" --durations=8 > $out/r05_shard.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_shard.out; tail -14 $out/r05_shard.out | cut -c1-250
[ $rc -ne 0 ] && exit $rc
bash tools/profile_round.sh r05 > $out/r05_profile.log 2>&1 || { tail -20 $out/r05_profile.log; exit 1; }
tail -2 $out/r05_profile.log | cut -c1-200
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_line.json ) 2> $out/r05_bench.time || exit 1
tail -3 $out/r05_bench.time
