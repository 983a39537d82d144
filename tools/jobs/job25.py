# Same-box A/B of the per-site epilogue forms of gp_eval: time the GP evaluation of the headline batch under site-kind tables that
# switch the round-2 forms off (3 -> 1: terminal-time form off; 4 -> 0: "u and div only" form off).
import ctypes as C, json, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from scasml_gp_amd import _lib
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
from scasml_gp_amd.solvers.ScaSML import ScaSML
d, n, B = 100, 3, 1 << 14
eq = Grad_Dependent_Nonlinear(d + 1)
np.random.seed(1234)
dom, bdy = eq.generate_data(1000, 200)
gp = GP_Grad_Dependent_Nonlinear(eq); gp.GPsolver(dom, bdy)
sol = ScaSML(eq, gp); eng = sol._engine
g = np.random.default_rng(1234)
x = torch.from_numpy(np.concatenate([g.uniform(-0.5, 0.5, (B, d)), g.uniform(0, 0.5, (B, 1))], axis=1).astype(np.float32)).cuda()
eng.solve(n, n, x)                      # fills the point buffer
pts, vals = eng._work["pts"], eng._work["vals"]
kinds = eng.site_kinds(n, n).clone()
ppr = kinds.numel()
tables = {"round-2 forms (3, 4 on)": kinds, "terminal-time form off (3 -> 1)": torch.where(kinds == 3, torch.ones_like(kinds), kinds),
          "u+div form off (4 -> 0)": torch.where(kinds == 4, torch.zeros_like(kinds), kinds),
          "round-1 forms (3 -> 1, 4 -> 0)": torch.where(kinds == 3, torch.ones_like(kinds), torch.where(kinds == 4, torch.zeros_like(kinds), kinds))}
out = {}
for rep in range(2):
    for name, k in tables.items():
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gp._eval_rows(pts, B * ppr, B, k, vals, x_bound=eng.path_bound()); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.setdefault(name, []).append(round(float(np.median(ts[2:])), 3))
print(json.dumps(out, indent=1))
