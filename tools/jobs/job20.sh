set -e
python tools/repeated_experiment.py --train-seeds 1234 1 2 3 > gpurun_out/r02_repeated_default.txt 2>&1
tail -22 gpurun_out/r02_repeated_default.txt | head -21
python tools/pde_loss_stats.py > gpurun_out/r02_pde_loss_stats.txt 2>&1
cat gpurun_out/r02_pde_loss_stats.txt | grep -v amdgpu
