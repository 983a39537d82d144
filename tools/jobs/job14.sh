set -e
timeout -k 10 1000 python tools/dist_gp_demo.py --ranks 2 --d 250 --n-dom 8333 --n-bdy 1667 > gpurun_out/r02_dist_gp_demo.json 2> gpurun_out/r02_dist_gp_demo.err || (tail -20 gpurun_out/r02_dist_gp_demo.err; exit 1)
cat gpurun_out/r02_dist_gp_demo.json
