#!/bin/bash
set -o pipefail
out=gpurun_out
python -u -m pytest tests -m gpu -x -q -k "shard or split or bench or rank or units or jax_stream or eight" --durations=5 > $out/r05_shard.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_shard.out; tail -10 $out/r05_shard.out | cut -c1-250
[ $rc -ne 0 ] && exit $rc
python tools/sample_sharding_times.py > $out/r05_sample_sharding_rank_times.txt 2>&1 && python tools/sample_sharding_times.py --variant fh --level 4 >> $out/r05_sample_sharding_rank_times.txt 2>&1
cat $out/r05_sample_sharding_rank_times.txt
bash tools/profile_round.sh r05 > $out/r05_profile.log 2>&1 || { tail -20 $out/r05_profile.log; exit 1; }
