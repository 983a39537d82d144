#!/usr/bin/env python
"""Development: where a distributed substitution (DistCholesky.solve) spends its time at M = 70 001 on one rank: wall time per solve with the collectives
skipped / forced through RCCL, and (under rocprofv3 --kernel-trace --stats) the kernels' own time.
    python tools/dist_solve_breakdown.py [--n-dom 16667 --n-bdy 3333] [--rccl]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=250)
ap.add_argument("--n-dom", type=int, default=16667)
ap.add_argument("--n-bdy", type=int, default=3333)
ap.add_argument("--rccl", action="store_true")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
if args.rccl:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from scasml_gp_amd.dist_gp import Comm, DistCholesky  # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear  # noqa: E402
from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear  # noqa: E402

eq = Grad_Dependent_Nonlinear(args.d + 1)
np.random.seed(1234)
dom, bdy = eq.generate_data(args.n_dom, args.n_bdy)
probe = GP_Grad_Dependent_Nonlinear(eq)
cm = Comm(force=args.rccl)
ch = DistCholesky(args.d, 1.0 / (0.25 ** 2 * args.d), dom, bdy, 1e-2, cm, compat_idx=probe.laplacian_idx).build().factor()
b = torch.from_numpy(np.random.default_rng(0).standard_normal(ch.M)).cuda()
ch.solve(b)
ch.matvec(b)
torch.cuda.synchronize()
for name, fn in (("solve", ch.solve), ("matvec", ch.matvec)):
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fn(b)
    t_issue = (time.perf_counter() - t0) / args.reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / args.reps
    print("%s: %.1f ms per call (host issue alone %.1f ms), collectives %s" % (name, t_all * 1e3, t_issue * 1e3, "through RCCL (one rank)" if args.rccl else "skipped (world = 1)"), flush=True)
if args.rccl:
    dist.destroy_process_group()
