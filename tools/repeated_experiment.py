#!/usr/bin/env python
"""The reference's RepeatedExperiment protocol (tests/RepeatedExperiment.py:50-141, 143-207) on the HIP
product path: GP trained on 1000+200 points, 10 repetitions with seeds 42..51 of 1000+200 test points,
n = rho = 2 (quadrature) or n = 2, M = 3 (full history); prints mean relative-L2 errors and mean solver
times next to the numbers the reference logged on an A800 (results*/**/RepeatedExperiment.log).

    python tools/repeated_experiment.py [--dims 20 40 60 80] [--reps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (GP, MLP, ScaSML) mean rel-L2 and mean seconds: results/Grad_Dependent_Nonlinear/{d}d/RepeatedExperiment/RepeatedExperiment.log:9,15,21,94,100,106
LOGGED_QUAD = {20: ((0.1456, 0.1576, 0.0690), (1.67, 27.6, 351.4)), 40: ((0.1844, 0.2104, 0.0950), (1.82, 31.3, 346.1)),
               60: ((0.2337, 0.2428, 0.1317), (1.51, 26.8, 333.1)), 80: ((0.2667, 0.2758, 0.1605), (1.61, 28.5, 378.2))}
# (MLP_fh, ScaSML_fh): results_full_history/.../RepeatedExperiment.log:15,21,100,106
LOGGED_FH = {20: ((0.1841, 0.0616), (1.22, 63.7)), 40: ((0.2269, 0.0891), (1.21, 55.5)),
             60: ((0.2517, 0.1234), (1.12, 56.0)), 80: ((0.2749, 0.1529), (1.23, 61.1))}


def rel_l2(sol, exact):       # tests/RepeatedExperiment.py:90-126
    sol, exact = np.asarray(sol, np.float64).ravel(), np.asarray(exact, np.float64).ravel()
    m = ~(np.isnan(sol) | np.isnan(exact))
    return float(np.linalg.norm(sol[m] - exact[m]) / np.linalg.norm(exact[m]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[20, 40, 60, 80])
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history

    rows = []
    for d in args.dims:
        np.random.seed(1234)                                   # experiment_run.py:32
        eq = Grad_Dependent_Nonlinear(d + 1)
        gp = GP_Grad_Dependent_Nonlinear(eq)
        dom, bdy = eq.generate_data(1000, 200)
        t0 = time.time()
        gp.GPsolver(dom, bdy, GN_steps=20)
        torch.cuda.synchronize()
        t_fit = time.time() - t0
        solvers = {"MLP": MLP(eq), "ScaSML": ScaSML(eq, gp), "MLP_fh": MLP_full_history(eq), "ScaSML_fh": ScaSML_full_history(eq, gp)}
        err = {k: [] for k in ["GP"] + list(solvers)}
        sec = {k: [] for k in ["GP"] + list(solvers)}
        for rep in range(args.reps):
            np.random.seed(42 + rep)                           # RepeatedExperiment.py:63-64, 200-207
            xt = np.concatenate(eq.generate_test_data(1000, 200), axis=0)
            exact = eq.exact_solution(xt)
            for name in err:
                torch.cuda.synchronize()
                t0 = time.time()
                if name == "GP":
                    sol = gp.predict(xt)
                elif name.endswith("_fh"):
                    sol = solvers[name].u_solve(2, None, xt, 3)
                else:
                    sol = solvers[name].u_solve(2, 2, xt)
                sec[name].append(time.time() - t0)
                err[name].append(rel_l2(sol, exact))
        row = {"d": d, "gp_fit_s": round(t_fit, 2)}
        for name in err:
            row[name] = {"rel_l2": round(float(np.mean(err[name])), 4), "std": round(float(np.std(err[name])), 4),
                         "ms": round(1e3 * float(np.mean(sec[name][1:] or sec[name])), 2)}
        q, f = LOGGED_QUAD.get(d), LOGGED_FH.get(d)
        if q:
            row["logged_rel_l2"] = dict(zip(["GP", "MLP", "ScaSML"], q[0]), MLP_fh=f[0][0], ScaSML_fh=f[0][1])
            row["logged_s"] = dict(zip(["GP", "MLP", "ScaSML"], q[1]), MLP_fh=f[1][0], ScaSML_fh=f[1][1])
        rows.append(row)
        print(json.dumps(row), flush=True)
    print("\n| d | solver | rel-L2 here (MI355X) | logged (A800) | time here | logged |\n|---|---|---|---|---|---|")
    for r in rows:
        for name in ["GP", "MLP", "ScaSML", "MLP_fh", "ScaSML_fh"]:
            lg = r.get("logged_rel_l2", {}).get(name, "-")
            ls = r.get("logged_s", {}).get(name, "-")
            print("| %d | %s | %.4f ± %.4f | %s | %.2f ms | %s s |" % (r["d"], name, r[name]["rel_l2"], r[name]["std"], lg, r[name]["ms"], ls))


if __name__ == "__main__":
    main()
