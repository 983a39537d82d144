#!/usr/bin/env python
"""Development: time the FP64-MFMA trailing-update tile (scasml_gemm_nt_sub: C -= A B^T, the kernel of the distributed
factorisation; gp_train.hip's chol_update_k_kernel is the same tile) on a square C of `n` rows at several panel depths K.
t(K) = t0 + c K separates the fixed cost of a tile (first operand fetch, the read-modify-write of C, the launch's tail) from its
matrix work.
    python tools/gemm_bench.py [n] [K ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from scasml_gp_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Ks = [int(a) for a in sys.argv[2:]] or [256, 512, 1024, 2048]
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
Kmax = max(Ks)
A = torch.rand((n, Kmax), dtype=torch.float64, device="cuda")
B = torch.rand((n, Kmax), dtype=torch.float64, device="cuda")
Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for K in Ks:
    ms = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.scasml_gemm_nt_sub(_lib.ptr(Cm), n, n, n, _lib.ptr(A), Kmax, _lib.ptr(B), Kmax, K, 0, 0, 0, s), "gemm_nt_sub")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    best = min(ms[1:])
    print("n=%d K=%d: %.3f ms (runs %s) = %.1f TFLOP/s FP64, %.2f us per 128 x 128 tile" % (
        n, K, best, ["%.3f" % m for m in ms], 2.0 * n * n * K / best / 1e9, best * 1e3 / ((n / 128) ** 2 / 256)), flush=True)
if os.environ.get("GEMM_BENCH_CHECK"):
    K = Ks[0]
    Cm.zero_()
    _lib.check(lib.scasml_gemm_nt_sub(_lib.ptr(Cm), n, n, n, _lib.ptr(A), Kmax, _lib.ptr(B), Kmax, K, 0, 0, 0, s), "gemm_nt_sub")
    ref = -(A[:2048, :K] @ B[:2048, :K].T)
    print("max |C - ref| on the first 2048 x 2048: %.3e" % float((Cm[:2048, :2048] - ref).abs().max()), flush=True)
