"""Development: evaluate one GP on one point set and dump the (n,4) outputs; run twice (SCASML_GP_OLD32 set / unset)
and compare the dumps:  python tools/dbg_eval_compare.py out.npy [d n_dom n_bdy n_points zero_every]"""
import sys
import numpy as np
import torch
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear

out = sys.argv[1]
d, nd, nb, n, zero_every = (int(a) for a in (sys.argv[2:7] + ["20", "60", "20", "15984", "3"][len(sys.argv) - 2:]))
rng = np.random.default_rng(6)
eq = Grad_Dependent_Nonlinear(d + 1)
dom, bdy = eq.generate_data(nd, nb) if False else (None, None)
dom = rng.uniform(-0.5, 0.5, (nd, d + 1)).astype(np.float16); dom[:, -1] = np.abs(dom[:, -1])
bdy = rng.uniform(-0.5, 0.5, (nb, d + 1)).astype(np.float16); bdy[:, -1] = 0.5
gp = GP_Grad_Dependent_Nonlinear(eq)
gp.GPsolver(dom, bdy, GN_steps=20)
x = rng.uniform(-0.5, 0.5, (n, d + 1)).astype(np.float32); x[:, -1] = np.abs(x[:, -1])
if zero_every:
    site = np.arange(n) // 24
    x[site % zero_every == 0] = 0.0
pts, _ = gp._points_device(x)
vals = gp._eval_device(pts)
torch.cuda.synchronize()
np.save(out, vals.cpu().numpy())
print("saved", out, vals.shape, float(vals.abs().max()))
