#!/usr/bin/env python
"""PDE residual of the trained surrogate on the harness test set, next to what the reference logged
(results/Grad_Dependent_Nonlinear/{d}d/SimpleUniform/SimpleUniform.log, line 'PDE Loss->')."""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear  # noqa: E402
from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear  # noqa: E402

LOGGED = {20: (-0.1092, 0.0773, -0.00270, 0.01565), 40: (-0.2209, 0.0876, -0.00370, 0.02433),   # min, max, mean, std
          60: (-0.1141, 0.0761, -0.00408, 0.02310), 80: (-0.1052, 0.0797, -0.00477, 0.02278)}    # {d}d/SimpleUniform/SimpleUniform.log:8
for compat in (False, True):
    print("surrogate:", "reference-compat (shifted 5-index Hutchinson, float16 entries), 4 index sets" if compat else "exact operators (default)")
    for d in (20, 40, 60, 80):
        rows = []
        for k in range(4 if compat else 1):
            np.random.seed(1234)
            eq = Grad_Dependent_Nonlinear(d + 1)
            idx = np.random.default_rng(2234 + k).choice(d, 5, replace=False)
            gp = GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=idx) if compat else GP_Grad_Dependent_Nonlinear(eq, compat=None)
            gp.GPsolver(*eq.generate_data(1000, 200), GN_steps=20)
            xt = np.concatenate(eq.generate_test_data(1000, 200))
            e = gp.compute_PDE_loss(xt)[:, 0].astype(np.float64)
            rows.append((e.min(), e.max(), e.mean(), e.std()))
        r = np.mean(rows, axis=0)
        print("d=%d PDE loss here: min %.4f max %.4f mean %.5f std %.5f   logged: %s" % (d, r[0], r[1], r[2], r[3], LOGGED.get(d)))
