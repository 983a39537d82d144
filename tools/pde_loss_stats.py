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
for d in (20, 40, 60, 80):
    np.random.seed(1234)
    eq = Grad_Dependent_Nonlinear(d + 1)
    gp = GP_Grad_Dependent_Nonlinear(eq)
    gp.GPsolver(*eq.generate_data(1000, 200), GN_steps=20)
    xt = np.concatenate(eq.generate_test_data(1000, 200))
    e = gp.compute_PDE_loss(xt)[:, 0].astype(np.float64)
    print("d=%d PDE loss here: min %.4f max %.4f mean %.5f std %.5f   logged: %s" % (d, e.min(), e.max(), e.mean(), e.std(), LOGGED.get(d)))
