#!/usr/bin/env python
"""CPU study (oracle only): the reference's RepeatedExperiment protocol (tests/RepeatedExperiment.py:50-141) on the
oracle, default modes vs reference-compat modes (oracle/gp_compat.py surrogate + compat_crn key reuse), next to the
logged means (results/**/RepeatedExperiment.log:9-22).

    python tests/studies/compat_solver_study.py --d 20 --train-seeds 1234 1 --reps 3
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
from oracle.mlp import PicardOracle                           # noqa: E402
from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear   # noqa: E402  (host-side sampler only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--train-seeds", type=int, nargs="+", default=[1234])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--modes", nargs="+", default=["default", "compat"])
    ap.add_argument("--variants", nargs="+", default=["quad"])
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
        dom, bdy = sampler.generate_data(1000, 200)
        dom, bdy = dom.astype(np.float64), bdy.astype(np.float64)
        idx = np.random.default_rng(1000 + ts).choice(d, 5, replace=False)
        for mode in args.modes:
            compat = mode == "compat"
            gp = OracleGPCompat(eq, idx) if compat else OracleGP(eq)
            gp.GPsolver(dom, bdy, GN_steps=20)
            for variant in args.variants:
                errs = {"GP": [], "MLP": [], "ScaSML": []}
                t0 = time.time()
                for r, (xt, ex) in enumerate(tests):
                    par = 2 if variant == "quad" else 3
                    errs["GP"].append(rel_l2(gp.predict(xt), ex))
                    errs["MLP"].append(rel_l2(PicardOracle(eq, variant, stream=r, compat_crn=compat).u_solve(2, par, xt), ex))
                    errs["ScaSML"].append(rel_l2(PicardOracle(eq, variant, gp=gp, stream=r, compat_crn=compat).u_solve(2, par, xt), ex))
                print(json.dumps({"d": d, "train_seed": ts, "mode": mode, "variant": variant, "idx": idx.tolist() if compat else None,
                                  **{k: round(float(np.mean(v)), 4) for k, v in errs.items()},
                                  "per_rep": {k: [round(e, 4) for e in v] for k, v in errs.items()}, "s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
