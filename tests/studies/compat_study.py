#!/usr/bin/env python
"""CPU study (oracle only): does the reference-compat surrogate (oracle/gp_compat.py) explain the gap between the
exact-operator GP of this build and the errors the reference logged?  Protocol of tests/RepeatedExperiment.py:
GP trained on 1000+200 float16 points, test sets of 1000+200 points (seeds 42..), relative L2 vs the exact solution.

    python tests/studies/compat_study.py --d 20 --train-seeds 1234 1 2 3 --idx-sets 4
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.equation import GradDependentNonlinear, rel_l2   # noqa: E402
from oracle.gp import OracleGP                                # noqa: E402
from oracle.gp_compat import OracleGPCompat                   # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear   # noqa: E402  (host-side sampler only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--train-seeds", type=int, nargs="+", default=[1234])
    ap.add_argument("--idx-sets", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--n-dom", type=int, default=1000)
    ap.add_argument("--n-bdy", type=int, default=200)
    ap.add_argument("--modes", nargs="+", default=["exact", "hutch", "hutch16"])
    args = ap.parse_args()
    d = args.d
    eq = GradDependentNonlinear(d + 1)
    sampler = Grad_Dependent_Nonlinear(d + 1)
    tests = []
    for r in range(args.reps):
        np.random.seed(42 + r)
        xt = np.concatenate(sampler.generate_test_data(1000, 200)).astype(np.float64)
        tests.append((xt, eq.exact_solution(xt)))
    for ts in args.train_seeds:
        np.random.seed(ts)
        dom, bdy = sampler.generate_data(args.n_dom, args.n_bdy)
        dom, bdy = dom.astype(np.float64), bdy.astype(np.float64)
        cases = []
        if "exact" in args.modes:
            cases.append(("exact", None))
        rng = np.random.default_rng(1000 + ts)
        for k in range(args.idx_sets):
            idx = rng.choice(d, 5, replace=False)
            if "hutch" in args.modes:
                cases.append(("hutch", idx))
            if "hutch16" in args.modes:
                cases.append(("hutch16", idx))
        for mode, idx in cases:
            t0 = time.time()
            gp = OracleGP(eq) if mode == "exact" else OracleGPCompat(eq, idx, round16=(mode == "hutch16"))
            gp.GPsolver(dom, bdy, GN_steps=20)
            errs = [rel_l2(gp.predict(xt), ex) for xt, ex in tests]
            print(json.dumps({"d": d, "train_seed": ts, "mode": mode, "idx": None if idx is None else idx.tolist(),
                              "gp_rel_l2": round(float(np.mean(errs)), 4), "per_rep": [round(e, 4) for e in errs],
                              "newton_steps": len(gp.loss_history) - 1, "loss": gp.loss_history[-1],
                              "K_eig_min": getattr(gp, "K_eig_min", None), "s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
