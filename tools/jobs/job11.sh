set -e
python -m pytest tests/test_gpu_dist_gp.py -m gpu -q -x --durations=5 > gpurun_out/r02_gputest10.log 2>&1 || (tail -40 gpurun_out/r02_gputest10.log; exit 1)
tail -8 gpurun_out/r02_gputest10.log
