python -X faulthandler -m pytest tests/test_gpu_rng.py -m gpu -q -x > gpurun_out/r02_rngtest.log 2>&1
echo "rc=$?"
grep -v "amdgpu.ids" gpurun_out/r02_rngtest.log | head -60 | cut -c1-250
