set -e
timeout -k 10 900 python tools/gp_large.py --d 250 --n-dom 16667 --n-bdy 3333 --roots 256 2>&1 | grep -v amdgpu.ids
