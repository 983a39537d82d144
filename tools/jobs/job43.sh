set -e
python tools/repeated_experiment.py --compat reference --train-seeds 1234 1 2 3 > gpurun_out/r02_repeated_compat.txt 2>&1
tail -3 gpurun_out/r02_repeated_compat.txt | cut -c1-300
python tools/repeated_experiment.py --train-seeds 1234 1 2 3 > gpurun_out/r02_repeated_default.txt 2>&1
tail -3 gpurun_out/r02_repeated_default.txt | cut -c1-300
