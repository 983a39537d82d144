"""File-based fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py with the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must hit them (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_rng_fixture():
    from oracle import philox
    g = _load("oracle_rng.npz")
    assert np.array_equal(philox.normals(0, 0, np.arange(64), 0, 20).view(np.uint32), g["normals_a"].view(np.uint32))
    assert np.array_equal(philox.normals(0xDEADBEEFCAFE, 9, np.arange(1 << 20, (1 << 20) + 64), 12345, 7).view(np.uint32),
                          g["normals_b"].view(np.uint32))
    assert np.array_equal(philox.uniform_tau(5, 2, np.arange(256), 77), g["tau"])


def test_oracle_reproduces_solver_and_gp_fixtures():
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.mlp import PicardOracle
    g = _load("oracle_mlp_d6.npz")
    eq = GradDependentNonlinear(7)
    for key in g.files:
        if key == "x_t":
            continue
        variant, n, par = key.split("_")
        got = PicardOracle(eq, variant, seed=3, stream=0).uz_solve(int(n), int(par), g["x_t"])
        assert np.allclose(got, g[key], rtol=0, atol=1e-12, equal_nan=True), key
    h = _load("oracle_gp_d6.npz")
    gp = OracleGP(eq)
    gp.GPsolver(h["x_dom"], h["x_bdy"], GN_steps=20)
    assert np.allclose(gp.right_vector, h["right_vector"], rtol=1e-9, atol=1e-9 * np.abs(h["right_vector"]).max())
    assert np.allclose(gp.predict(h["X"]), h["predict"], atol=1e-10)
    assert np.allclose(PicardOracle(eq, "quad", gp=gp, seed=3, stream=0).uz_solve(2, 2, h["x_t"]), h["scasml_quad_2_2"], atol=1e-10)


@pytest.mark.gpu
def test_hip_hits_the_fixtures():
    import torch
    from scasml_gp_amd import _lib
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    lib = _lib.load()
    g = _load("oracle_rng.npz")
    out = torch.empty((64, 20), dtype=torch.float32, device="cuda")
    _lib.check(lib.scasml_debug_normals(_lib.Rng(0, 0, 0, 0, 1, 0, 0), 0, 20, 64, _lib.ptr(out), _lib.stream_ptr()), "normals")
    assert np.array_equal(out.cpu().numpy().view(np.uint32), g["normals_a"].view(np.uint32))
    m = _load("oracle_mlp_d6.npz")
    eq = Grad_Dependent_Nonlinear(7)
    for key in m.files:
        if key == "x_t":
            continue
        variant, n, par = key.split("_")
        got = MLP(eq, seed=3).uz_solve(int(n), int(par), m["x_t"]) if variant == "quad" \
            else MLP_full_history(eq, seed=3).uz_solve(int(n), None, m["x_t"], int(par))
        want = m[key]
        assert np.array_equal(np.isnan(got), np.isnan(want)), key
        ok = ~np.isnan(want)
        assert np.all(np.abs(got[ok] - want[ok]) <= 2e-5 + 1e-4 * np.abs(want[ok])), key
    h = _load("oracle_gp_d6.npz")
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    gp.GPsolver(h["x_dom"], h["x_bdy"], GN_steps=20)
    assert np.abs(gp.right_vector - h["right_vector"]).max() <= 1e-7 * np.abs(h["right_vector"]).max()
    assert np.allclose(gp.loss_history, h["loss_history"], rtol=1e-8)
    scale = 1e-5 * (np.abs(h["right_vector"]).sum() + 1)
    assert np.abs(gp.predict(h["X"]) - h["predict"]).max() <= scale
    assert np.abs(gp.compute_gradient(h["X"]) - h["gradient"]).max() <= 20 * scale
    assert np.abs(gp.compute_PDE_loss(h["X"]) - h["pde"]).max() <= 20 * scale
    for cls, key, args in ((ScaSML, "scasml_quad_2_2", (2, 2, h["x_t"])), (ScaSML_full_history, "scasml_fh_2_3", (2, None, h["x_t"], 3))):
        got = cls(eq, gp, seed=3).uz_solve(*args)
        assert np.all(np.abs(got - h[key]) <= 5e-5 + 2e-4 * np.abs(h[key])), key
