#!/bin/bash
# round-5 GPU batch 2: distributed-GP tests on the new substitutions, rocprofv3 passes of the bench command (both evaluation modes), the driver's
# bench command with its wall time, the M = 70 001 distributed fit under RCCL again
set -o pipefail
out=gpurun_out
python -u -m pytest tests/test_gpu_dist_gp.py tests/test_gpu_xl.py -x -q --durations=8 > $out/r05_dist.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_dist.out; tail -14 $out/r05_dist.out
[ $rc -ne 0 ] && exit $rc
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_line_driver_cmd.json ) 2> $out/r05_bench_driver_cmd.time || exit 1
tail -4 $out/r05_bench_driver_cmd.time
bash tools/profile_round.sh r05 > $out/r05_profile.log 2>&1 || { tail -20 $out/r05_profile.log; exit 1; }
tail -3 $out/r05_profile.log | cut -c1-300
bash tools/profile_round.sh r05geo --compat reference-geometry > $out/r05geo_profile.log 2>&1 || { tail -20 $out/r05geo_profile.log; exit 1; }
tail -3 $out/r05geo_profile.log | cut -c1-300
python tools/dist_gp_demo.py --ranks 1 --backend nccl --n-dom 16667 --n-bdy 3333 > $out/r05_dist_gp_70k_rccl_one_rank_v2.json 2> $out/r05_dist_gp_70k_v2.err || { tail -20 $out/r05_dist_gp_70k_v2.err; exit 1; }
grep '^{' $out/r05_dist_gp_70k_rccl_one_rank_v2.json | cut -c1-900
