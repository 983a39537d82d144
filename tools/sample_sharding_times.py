#!/usr/bin/env python
"""Development: what each rank of a Monte-Carlo sample-sharded step costs, measured on ONE GPU by running the ranks' shares in turn (HIP events,
best of 5): the compute side of the strong-scaling leg bench.py times on a multi-GPU node (the all-reduce of (B, 1+d) floats is not in here).
    python tools/sample_sharding_times.py [--variant quad|fh] [--level 3] [--worlds 2 4 8] [--breakdown] [--kinds]
--breakdown: per rank also the HIP-event time of each kernel, the host's issue time of a step (wall clock without a device synchronise) and the
             number of sites the rank evaluates, by kind -- where the ranks' times sum to more than the unsharded step
--kinds:     the GP evaluation alone on the whole point buffer with every site declared of ONE kind: the per-site cost by kind that the dealing's
             cost model (scasml_plan_deal_units) should charge"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from scasml_gp_amd import _lib  # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="quad")
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--M", type=int, default=3)
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--breakdown", action="store_true")
ap.add_argument("--kinds", action="store_true")
ap.add_argument("--compat", default="reference")
args = ap.parse_args()
d, B = 100, 1 << 14
eq = Grad_Dependent_Nonlinear(d + 1)
eq.geometry()
x_dom, x_bdy, _ = bench.harness_sets(eq, 1000, 200)
gp, _ = bench.fit_surrogate(eq, x_dom, x_bdy, args.compat)
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


def steps_ms(fn, k=10, reps=3):
    """ms per step of k steps issued back to back (no synchronise between them: what bench.py's timed region does), best of `reps`."""
    fn()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / k)
    return min(out)


def kernel_breakdown(fn, reps=5):
    """Per-kernel HIP-event averages of `reps` steps and the host's issue time per step (no synchronise inside the loop)."""
    fn()
    torch.cuda.synchronize()
    eng.profile = True
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps * 1e3
    kms = eng.collect_kernel_ms()
    eng.profile = False
    return {k: round(v, 3) for k, v in kms.items()}, round(host, 3)


def kind_counts(kinds):
    k = kinds.cpu().numpy()
    return {"full(0)": int((k == 0).sum()), "u_div(4)": int((k == 4).sum()), "terminal(3)": int((k == 3).sum()), "root(1)": int((k == 1).sum()),
            "not_mine(2)": int((k == 2).sum())}


full_single = best_ms(lambda: eng.solve(wl.n, wl.par, wl.x_dev, stream_id=1))
full = steps_ms(lambda: eng.solve(wl.n, wl.par, wl.x_dev, stream_id=1), k=5)
rec = {"workload": wl.name, "roots": B, "unsharded_ms": round(full, 3), "unsharded_ms_single_step_then_synchronise": round(full_single, 3)}
if args.breakdown:
    rec["kernel_ms"], rec["host_issue_ms"] = kernel_breakdown(lambda: eng.solve(wl.n, wl.par, wl.x_dev, stream_id=1))
    rec["sites"] = kind_counts(eng.site_kinds(wl.n, wl.par))
print(json.dumps(rec), flush=True)

if args.kinds:
    # the evaluation alone, every site of one kind: ms per launch over the buffer the step above left behind, and per (site x 16384 roots)
    lib = _lib.load()
    plan = eng.plan(wl.n, wl.par)
    ppr = int(lib.scasml_points_per_root(C.byref(plan)))
    kp = int(lib.scasml_point_stride(d))
    stride = (B + 31) // 32 * 32
    pts, vals = eng._buffers(stride * ppr, kp)
    xb = eng.path_bound(0.5, plan)
    for kind, name in ((0, "full(0)"), (4, "u_div(4)"), (3, "terminal(3)"), (1, "root(1)")):
        kinds = torch.full((ppr,), kind, dtype=torch.uint8, device="cuda")
        ms = best_ms(lambda: gp._eval_rows(pts, stride * ppr, stride, kinds, vals, x_bound=xb))
        print(json.dumps({"gp_eval_all_sites_of_kind": name, "ms": round(ms, 3), "us_per_site": round(ms / ppr * 1e3, 2)}), flush=True)
    kinds = torch.full((ppr,), 2, dtype=torch.uint8, device="cuda")
    ms = best_ms(lambda: gp._eval_rows(pts, stride * ppr, stride, kinds, vals, x_bound=xb))
    print(json.dumps({"gp_eval_all_sites_of_kind": "not_mine(2): every workgroup returns at once", "ms": round(ms, 3), "workgroups": stride * ppr // 128}), flush=True)

for world in args.worlds:
    load = eng.unit_owners(wl.n, wl.par, world)[2]
    times = [steps_ms(lambda r=r: eng.solve(wl.n, wl.par, wl.x_dev, rank=r, world=world, stream_id=1)) for r in range(world)]
    single = [best_ms(lambda r=r: eng.solve(wl.n, wl.par, wl.x_dev, rank=r, world=world, stream_id=1)) for r in range(world)]
    whole = float(eng.unit_owners(wl.n, wl.par, 1)[2][0])
    rec = {"sample_ranks": world, "dealt_load_max_over_mean": round(float(load.max() / load.mean()), 3),
           "dealt_load_sum_over_unsharded": round(float(load.sum()) / whole, 3),
           "rank_ms": [round(t, 3) for t in times], "slowest_rank_ms": round(max(times), 3), "sum_of_ranks_over_unsharded": round(sum(times) / full, 3),
           "compute_speedup": round(full / max(times), 2), "compute_efficiency": round(full / max(times) / world, 3),
           "modelled_slowest_rank_ms": round(full * float(load.max()) / whole + bench.SAMPLE_RANK_FIXED_MS, 3),
           "single_step_then_synchronise": {"rank_ms": [round(t, 3) for t in single], "compute_efficiency": round(full_single / max(single) / world, 3)}}
    print(json.dumps(rec), flush=True)
    if args.breakdown:
        for r in range(world):
            kms, host = kernel_breakdown(lambda r=r: eng.solve(wl.n, wl.par, wl.x_dev, rank=r, world=world, stream_id=1))
            print(json.dumps({"sample_ranks": world, "rank": r, "dealt_load": round(float(load[r]), 2), "kernel_ms": kms, "host_issue_ms": host,
                              "sites": kind_counts(eng.site_kinds(wl.n, wl.par, r, world))}), flush=True)
