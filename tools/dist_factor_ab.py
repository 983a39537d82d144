#!/usr/bin/env python
"""Development: DistCholesky.factor one block column at a time (pair=False, round 5) against two at a time with one K = 512 trailing update
(pair=True) on one rank -- wall time of the factorisation and the two factors against each other.
    python tools/dist_factor_ab.py [--n-dom 16667 --n-bdy 3333] [--rccl]"""
import argparse
import json
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
args = ap.parse_args()
if args.rccl:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
out = {"M": 4 * args.n_dom + args.n_bdy, "rccl_one_rank": bool(args.rccl)}
sample = None
for pair in (False, True, False, True):
    ch = DistCholesky(args.d, 1.0 / (0.25 ** 2 * args.d), dom, bdy, 1e-2, cm, compat_idx=probe.laplacian_idx).build()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ch.factor(pair=pair)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    key = "pair" if pair else "single"
    out.setdefault(key + "_factor_s", []).append(round(dt, 3))
    out[key + "_tflops"] = round(ch.M ** 3 / 3 / min(out[key + "_factor_s"]) / 1e12, 2)
    rows = ch.R[-512:, :].clone()           # the last two block rows of the factor: every update has touched them
    if sample is None:
        sample = rows
    elif pair:
        out["max_rel_diff_of_the_last_block_rows"] = float((rows - sample).abs().max() / sample.abs().max())
    del ch, rows
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
