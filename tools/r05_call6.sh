#!/bin/bash
set -o pipefail
out=gpurun_out
python -u -m pytest tests/test_gpu_dist_gp.py::test_inexact_newton_reaches_the_same_fit_in_fewer_products tests/test_gpu_scasml.py::test_root_bound_is_taken_per_solve_for_converted_inputs tests/test_gpu_full_size.py::test_config1_mlp_d20_n2_at_its_full_batch_of_2_20_roots -x -q -s > $out/r05_new_tests.out 2>&1; rc=$?; echo "rc=$rc" >> $out/r05_new_tests.out; tail -15 $out/r05_new_tests.out | cut -c1-400
[ $rc -ne 0 ] && exit $rc
python tools/dist_gp_demo.py --ranks 1 --backend nccl --n-dom 16667 --n-bdy 3333 --adaptive-cg > $out/r05_dist_gp_70k_rccl_one_rank_adaptive.json 2> $out/r05_dist_gp_70k_adaptive.err || { tail -20 $out/r05_dist_gp_70k_adaptive.err; exit 1; }
grep '^{' $out/r05_dist_gp_70k_rccl_one_rank_adaptive.json | cut -c1-1500
