#!/usr/bin/env python
"""Development: time scasml_cholesky alone on a synthetic SPD matrix (run time does not depend on the values).
    python tools/chol_bench.py [M] [reps] [inverse]       prints ms and TFLOP/s (M^3/3 flops; with `inverse` also K^-1 from the factor, 2 M^3/3)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from scasml_gp_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 35008
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
A0 = torch.rand((M, M), dtype=torch.float64, device="cuda") * 1e-3
A0 = A0 + torch.eye(M, dtype=torch.float64, device="cuda") * 2.0
A = torch.empty_like(A0)
info = torch.zeros(1, dtype=torch.int32, device="cuda")
ms = []
for r in range(reps + 1):
    A.copy_(A0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.scasml_cholesky(_lib.ptr(A), M, 0.0, _lib.ptr(info), s), "cholesky")
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
best = min(ms[1:]) if reps else ms[0]      # reps = 0: ONE cold factorisation (counter passes)
print("M=%d cholesky %.1f ms (runs %s) = %.1f TFLOP/s FP64" % (M, best, ["%.1f" % m for m in (ms[1:] if reps else ms)], M ** 3 / 3 / best / 1e9), flush=True)
assert int(info.item()) == 0 or os.environ.get("SCASML_HIP_LIB"), "factorisation failed"
if len(sys.argv) > 3 and sys.argv[3] == "inverse":
    Ainv = A0          # the input is no longer needed
    ms = []
    for r in range(max(reps, 1) + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(A), M, _lib.ptr(Ainv), s), "cholesky_inverse")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    best = min(ms[1:])
    print("M=%d inverse from the factor %.1f ms (runs %s) = %.1f TFLOP/s FP64" % (M, best, ["%.1f" % m for m in ms[1:]], 2 * M ** 3 / 3 / best / 1e9), flush=True)
