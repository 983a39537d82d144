#!/usr/bin/env python
"""Development: does a HIP graph shorten the launch-bound chain of the small-M factorisation?  scasml_cholesky (+ the two triangular solves of a
Newton step) captured once on a side stream (torch.cuda.graph = hipStreamBeginCapture) and replayed, against the same calls issued directly.
    python tools/chol_graph_bench.py [M] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from scasml_gp_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4224
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = _lib.load()
A0 = torch.rand((M, M), dtype=torch.float64, device="cuda") * 1e-3
A0 = A0 + A0.T + torch.eye(M, dtype=torch.float64, device="cuda") * 2.0
A = torch.empty_like(A0)
b0 = torch.rand((M, 1), dtype=torch.float64, device="cuda")
b = torch.empty_like(b0)
info = torch.zeros(1, dtype=torch.int32, device="cuda")


def work():
    s = _lib.stream_ptr()
    A.copy_(A0)
    b.copy_(b0)
    _lib.check(lib.scasml_cholesky(_lib.ptr(A), M, 1e-4, _lib.ptr(info), s), "cholesky")
    _lib.check(lib.scasml_trsm_lower(_lib.ptr(A), M, _lib.ptr(b), 1, 0, s), "trsm")
    _lib.check(lib.scasml_trsm_lower(_lib.ptr(A), M, _lib.ptr(b), 1, 1, s), "trsm^T")


def timed(fn):
    ms = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms[1:]), sorted(ms[1:])[len(ms[1:]) // 2]


work()
torch.cuda.synchronize()
ref = b.clone()
direct = timed(work)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    work()
torch.cuda.synchronize()
b.zero_()
g.replay()
torch.cuda.synchronize()
same = bool(torch.equal(b, ref))
graph = timed(g.replay)
print("M=%d factor + 2 substitutions: direct %.3f ms (median %.3f), graph replay %.3f ms (median %.3f), identical result %s, info %d"
      % (M, direct[0], direct[1], graph[0], graph[1], same, int(info.item())), flush=True)
