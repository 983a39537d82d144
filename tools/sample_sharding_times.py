#!/usr/bin/env python
"""Development: what each rank of a Monte-Carlo sample-sharded step costs, measured on ONE GPU by running the ranks' shares in turn (HIP events,
best of 5): the compute side of the strong-scaling leg bench.py times on a multi-GPU node (the all-reduce of (B, 1+d) floats is not in here).
    python tools/sample_sharding_times.py [--variant quad|fh] [--level 3] [--worlds 2 4 8]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="quad")
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--M", type=int, default=3)
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
args = ap.parse_args()
d, B = 100, 1 << 14
eq = Grad_Dependent_Nonlinear(d + 1)
eq.geometry()
x_dom, x_bdy, _ = bench.harness_sets(eq, 1000, 200)
gp, _ = bench.fit_surrogate(eq, x_dom, x_bdy, "reference")
wl = bench.Workload(eq, gp, "scasml", args.variant, args.level, args.M, B, 0)
eng = wl.eng


def best_ms(fn, reps=5):
    fn()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return min(out)


full = best_ms(lambda: eng.solve(wl.n, wl.par, wl.x_dev, stream_id=1))
print(json.dumps({"workload": wl.name, "roots": B, "unsharded_ms": round(full, 3)}), flush=True)
for world in args.worlds:
    load = eng.unit_owners(wl.n, wl.par, world)[2]
    times = [best_ms(lambda r=r: eng.solve(wl.n, wl.par, wl.x_dev, rank=r, world=world, stream_id=1)) for r in range(world)]
    print(json.dumps({"sample_ranks": world, "dealt_load_max_over_mean": round(float(load.max() / load.mean()), 3),
                      "rank_ms": [round(t, 3) for t in times], "slowest_rank_ms": round(max(times), 3),
                      "compute_speedup": round(full / max(times), 2), "compute_efficiency": round(full / max(times) / world, 3)}), flush=True)
