#!/usr/bin/env python
"""Regenerates tests/golden/oracle_*.npz with the CPU oracle (oracle/).

The reference cannot be run here (jax / deepxde are not installed) and ships no golden vectors, so
these fixtures freeze the ORACLE's outputs on fixed seeded inputs: they guard the restatement against
silent drift (CPU test) and give the HIP path a file-based target (GPU test).  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import philox  # noqa: E402
from oracle.equation import GradDependentNonlinear, sample_points  # noqa: E402
from oracle.gp import OracleGP  # noqa: E402
from oracle.mlp import PicardOracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # 1. RNG: normals of two sites and the full-history uniform
    np.savez_compressed(os.path.join(HERE, "oracle_rng.npz"),
                        normals_a=philox.normals(0, 0, np.arange(64), 0, 20),
                        normals_b=philox.normals(0xDEADBEEFCAFE, 9, np.arange(1 << 20, (1 << 20) + 64), 12345, 7),
                        tau=philox.uniform_tau(5, 2, np.arange(256), 77))
    # 2. plain MLP, quadrature and full history
    d = 6
    eq = GradDependentNonlinear(d + 1)
    xt = np.concatenate(sample_points(np.random.default_rng(123), d, 12, 4)).astype(np.float32)
    out = {"x_t": xt}
    for n, rho in ((1, 1), (2, 2), (3, 3), (2, 4)):
        out["quad_%d_%d" % (n, rho)] = PicardOracle(eq, "quad", seed=3, stream=0).uz_solve(n, rho, xt)
    for n, M in ((2, 3), (3, 2)):
        out["fh_%d_%d" % (n, M)] = PicardOracle(eq, "fh", seed=3, stream=0).uz_solve(n, M, xt)
    np.savez_compressed(os.path.join(HERE, "oracle_mlp_d6.npz"), **out)
    # 3. GP fit + posterior + ScaSML on the defect
    dom, bdy = sample_points(np.random.default_rng(5), d, 40, 12)
    gp = OracleGP(eq)
    gp.GPsolver(dom, bdy, GN_steps=20)
    X = np.random.default_rng(6).uniform(-0.5, 0.5, (24, d + 1)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1])
    dt, div, lap = gp.pde_parts(X)
    np.savez_compressed(os.path.join(HERE, "oracle_gp_d6.npz"), x_dom=dom, x_bdy=bdy, right_vector=gp.right_vector,
                        loss_history=np.array(gp.loss_history), X=X, predict=gp.predict(X), gradient=gp.compute_gradient(X),
                        dt=dt, div=div, lap=lap, pde=gp.compute_PDE_loss(X), x_t=xt,
                        scasml_quad_2_2=PicardOracle(eq, "quad", gp=gp, seed=3, stream=0).uz_solve(2, 2, xt),
                        scasml_fh_2_3=PicardOracle(eq, "fh", gp=gp, seed=3, stream=0).uz_solve(2, 3, xt))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
