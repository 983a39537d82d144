set -e
python -m pytest tests -m gpu -q > gpurun_out/r02_gputest_final.log 2>&1 || (tail -40 gpurun_out/r02_gputest_final.log; exit 1)
tail -3 gpurun_out/r02_gputest_final.log
bash tools/profile_round.sh r02
