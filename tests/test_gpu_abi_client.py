"""The C ABI is usable without Python: tests/abi_client.cpp (hipcc, links libscasml_hip.so only) must produce the
same numbers, bit for bit, as the Python class path for the same seed and inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_client_matches_python_path(tmp_path):
    from scasml_gp_amd import _build
    lib_dir = os.path.dirname(_build.build_library())
    exe = str(tmp_path / "abi_client")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_client.cpp"),
                    "-L", lib_dir, "-lscasml_hip", "-Wl,-rpath," + lib_dir, "-o", exe], check=True, capture_output=True)
    d, B, seed = 10, 8, 5
    r = subprocess.run([exe, str(d), str(B), str(seed)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.array([float(v) for v in r.stdout.split()], dtype=np.float32)
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    x = np.zeros((B, d + 1), dtype=np.float32)
    for b in range(B):
        for k in range(d):
            x[b, k] = np.float32(-0.5) + np.float32((b * 131 + k * 17) % 97) / np.float32(96.0)
        x[b, d] = np.float32(0.5) * np.float32((b * 29) % 50) / np.float32(50.0)
    want = MLP_full_history(Grad_Dependent_Nonlinear(d + 1), seed=seed).u_solve(2, None, x, 3)[:, 0]
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
