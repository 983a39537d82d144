#!/usr/bin/env python
"""BASELINE configs[4] staged, the SOLVER half alone: ScaSML n = rho = 3 at d = 250 on the as-coded surrogate of 16 667 + 3 333 collocation points
(M = 70 001; the packed model is 34 MB: past the 4 MB L2 of an XCD, inside the 256 MB Infinity Cache), 1024 roots.
    python tools/xl_solver_bench.py [--state gpurun_out/xl_state.npz] [--steps 5] [--roots 1024]
The first run fits the surrogate (about 35 s) and saves its state; later runs (the rocprofv3 counter passes of tools/xl_solver_counters.sh) load it.
Prints one JSON line: ms per step, HIP-event kernel times, and the roofline of gp_eval_compat_mfma_kernel<16, ...> at this shape."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear  # noqa: E402
from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--state", default=os.path.join("gpurun_out", "xl_state.npz"))
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--roots", type=int, default=1024)
args = ap.parse_args()
d, nd, nb = 250, 16667, 3333
eq = Grad_Dependent_Nonlinear(d + 1)
eq.geometry()
gp = GP_Grad_Dependent_Nonlinear(eq)
fit_s = None
if os.path.exists(args.state):
    gp.load(args.state)
else:
    import time
    dom, bdy, _ = bench.harness_sets(eq, nd, nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.GPsolver(dom, bdy, GN_steps=20)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    gp._L_pad = gp.cholesky_phi_phi_perturb = None
    torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(os.path.abspath(args.state)), exist_ok=True)
    gp.save(args.state)

class _Ranks:
    on, rank, world = False, 0, 1

    def max_seconds(self, dt):
        return dt

wl = bench.Workload(eq, gp, "scasml", "quad", 3, 3, args.roots, 0)
elapsed, kms = bench.measure(_Ranks(), wl, wl.step, args.steps, 2)
n_inf = wl.B * (wl.steps_exec + 1)
n_colloc, m_feat = nd + nb, 4 * nd + nb
flops = n_inf * (2.0 * n_colloc * (d + 1) + 10.0 * m_feat)
tf = flops / (kms["gp_eval"] * 1e-3) / 1e12
model_mb = gp._compat_model.numel() * 4 / 1e6
print(json.dumps({"workload": "Grad_Dependent_Nonlinear d=250, %s, B=%d roots, as-coded surrogate of %d+%d collocation points (M = %d): BASELINE configs[4] staged, solver half"
                              % (wl.name, wl.B, nd, nb, m_feat),
                  "fit_s": round(fit_s, 2) if fit_s else None, "steps": args.steps, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                  "value": round(wl.B * wl.steps_exec * args.steps / elapsed, 1), "unit": "path-steps/s (executed)",
                  "kernel_ms": {k: round(v, 4) for k, v in kms.items()}, "n_inf": n_inf, "packed_model_mb": round(model_mb, 1),
                  "roofline": {"kernel": "gp_eval_compat_mfma_kernel<16, 2, true, 2>", "bound": "valu+mfma (sum model)", "flops_per_launch": flops,
                               "achieved": round(tf, 1), "peak": bench.MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / bench.MFMA_BF16_PEAK_TFLOPS, 4),
                               "note": "algorithmic flops of SURVEY 8(d), 2 N_inf N (d+1) + 10 N_inf M; counters: profiles/r06_gp_eval_pmc_d250.json"}}), flush=True)
