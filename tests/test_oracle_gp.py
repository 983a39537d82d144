"""Oracle GP: closed-form derivative kernels vs finite differences of kappa, Newton
gradient/Hessian vs finite differences of the loss, posterior derivatives vs finite
differences of the posterior mean (replaces the commented-out self-checks at
models/GP.py:446-485)."""
import numpy as np
import pytest

from oracle.equation import GradDependentNonlinear, sample_points
from oracle.gp import OracleGP


def _kappa(gp, x, y):
    return np.exp(-np.sum((x - y) ** 2) / (2 * gp.s2))


def _apply(op, fun, x, h):
    """Finite-difference operator on the first argument of fun (time is the LAST coordinate)."""
    d = len(x) - 1
    if op == "I":
        return fun(x)
    if op == "dt":
        e = np.zeros(d + 1); e[-1] = h
        return (fun(x + e) - fun(x - e)) / (2 * h)
    if op == "div":
        e = np.zeros(d + 1); e[:d] = h          # directional derivative along (1,..,1,0) == sum of partials
        return (fun(x + e) - fun(x - e)) / (2 * h)
    if op == "lap":
        s = 0.0
        f0 = fun(x)
        for i in range(d):
            e = np.zeros(d + 1); e[i] = h
            s += (fun(x + e) - 2 * f0 + fun(x - e)) / (h * h)
        return s
    raise KeyError(op)


@pytest.mark.parametrize("d", [3, 10])
def test_closed_form_blocks_match_finite_differences(d):
    eq = GradDependentNonlinear(d + 1)
    gp = OracleGP(eq)
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, d + 1)
    y = rng.uniform(-0.5, 0.5, d + 1)
    ops = ["I", "lap", "dt", "div"]
    for ox in ops:
        for oy in ops:
            hx = 2e-2 if "lap" in (ox, oy) else 1e-3
            inner = lambda xx: _apply(oy, lambda yy: _kappa(gp, xx, yy), y, hx)
            fd = _apply(ox, inner, x, hx)
            cf = gp.block(ox, oy, x[None, :], y[None, :])[0, 0]
            scale = max(1.0, abs(cf))
            tol = 2e-2 if (ox == "lap" and oy == "lap") else 5e-3 if "lap" in (ox, oy) else 1e-5
            assert abs(fd - cf) < tol * scale * gp.a ** 2, (ox, oy, fd, cf)


def test_gram_is_symmetric_and_block_ordered():
    d = 5
    eq = GradDependentNonlinear(d + 1)
    gp = OracleGP(eq)
    dom, bdy = sample_points(np.random.default_rng(1), d, 12, 5)
    K = gp.kernel_phi_phi(dom, bdy)
    assert K.shape == (4 * 12 + 5, 4 * 12 + 5)
    assert np.allclose(K, K.T, atol=1e-12)
    assert np.allclose(np.diag(K)[:17], 1.0)           # kappa(x, x) = 1 on the u(X) rows
    assert np.linalg.eigvalsh(K).min() > -1e-9          # PSD


def _fit_small(d=4, nd=30, nb=10, steps=20):
    eq = GradDependentNonlinear(d + 1)
    gp = OracleGP(eq)
    dom, bdy = sample_points(np.random.default_rng(2), d, nd, nb)
    gp.GPsolver(dom, bdy, GN_steps=steps)
    return eq, gp, dom, bdy


def test_newton_decreases_loss_and_reaches_stationarity():
    eq, gp, dom, bdy = _fit_small()
    h = gp.loss_history
    assert h[-1] < h[0]
    # stationarity: finite-difference gradient of the loss at the solution is ~0
    K = gp.kernel_phi_phi(dom, bdy)
    A = np.linalg.inv(K + gp.nugget * np.eye(K.shape[0]))
    g = eq.g(gp.x_t_boundary)[:, 0]
    loss = lambda s: gp._b(s, g) @ A @ gp._b(s, g)
    fd = np.array([(loss(gp.sol + 1e-6 * e) - loss(gp.sol - 1e-6 * e)) / 2e-6 for e in np.eye(len(gp.sol))[:10]])
    assert np.abs(fd).max() < 1e-4


def test_posterior_derivatives_match_finite_differences():
    eq, gp, dom, bdy = _fit_small()
    d = eq.d
    X = np.random.default_rng(3).uniform(-0.4, 0.4, (6, d + 1)); X[:, -1] = np.abs(X[:, -1])
    grad = gp.compute_gradient(X)
    dt, div, lap = gp.pde_parts(X)
    h = 1e-5
    for i in range(d + 1):
        e = np.zeros(d + 1); e[i] = h
        fd = (gp.predict(X + e) - gp.predict(X - e))[:, 0] / (2 * h)
        assert np.allclose(fd, grad[:, i], atol=1e-6, rtol=1e-5)
    assert np.allclose(grad[:, -1], dt[:, 0], atol=1e-10)
    assert np.allclose(grad[:, :-1].sum(1), div[:, 0], atol=1e-10)
    h = 1e-3
    fl = np.zeros(len(X))
    for i in range(d):
        e = np.zeros(d + 1); e[i] = h
        fl += (gp.predict(X + e) - 2 * gp.predict(X) + gp.predict(X - e))[:, 0] / (h * h)
    assert np.allclose(fl, lap[:, 0], atol=1e-4, rtol=1e-4)


def test_gp_interpolates_boundary_data_and_reduces_pde_residual():
    eq, gp, dom, bdy = _fit_small(d=4, nd=60, nb=30)
    # with nugget 1e-2 the fit is a smoother, not an interpolant; it must still track g on the boundary set
    err = np.abs(gp.predict(gp.x_t_boundary) - eq.g(gp.x_t_boundary)).mean()
    assert err < 0.05
    assert np.abs(gp.compute_PDE_loss(gp.x_t_domain)).mean() < 0.1
