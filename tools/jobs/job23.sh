set -e
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r02_gputest_final.log 2>&1 || (tail -40 gpurun_out/r02_gputest_final.log; exit 1)
tail -12 gpurun_out/r02_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()"
