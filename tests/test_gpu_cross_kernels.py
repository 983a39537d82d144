"""The cross-kernel builders of the reference's GP class surface (models/GP.py:41-179, 271-411, 630-651) as host views over
scasml_gp_cross_rows: kernel_x_t_phi, laplacian_ / dt_ / div_x_t_kernel_x_t_phi, dx_t_kernel_x_t_phi, kernel_x_t_phi_single, kappa and the
derivative kernels of single pairs -- against the oracle's blocks (oracle/gp_compat.py for the surrogate as coded, oracle/gp.py for the
documented operators).  VERDICT r5 item 9: a caller of these methods used to get AttributeError."""
import numpy as np
import pytest

from test_gpu_compat_mfma import _setup, _test_points

pytestmark = pytest.mark.gpu

OPS = ("I", "lap", "dt", "div")


def _f16_ulp(v):
    return np.maximum(np.abs(v), 2.0 ** -14) * 2.0 ** -10


@pytest.mark.parametrize("d,idx,nd,nb", [(20, [11, 17, 12, 6, 4], 60, 21), (7, [5, 0, 3, 6, 2], 33, 0), (100, [99, 0, 50, 7, 31], 40, 17)])
def test_as_coded_feature_rows_are_the_oracles_float16_entries(d, idx, nd, nb):
    gp, ogp, _ = _setup(d, idx, nd, max(nb, 1), seed=31)
    dom, bdy = gp.x_t_domain, gp.x_t_boundary[:nb]
    ogp.x_t_boundary, ogp.N_boundary = ogp.x_t_boundary[:nb], nb
    X = _test_points(d, 77, seed=32)
    views = {"I": gp.kernel_x_t_phi, "lap": gp.laplacian_x_t_kernel_x_t_phi, "dt": gp.dt_x_t_kernel_x_t_phi, "div": gp.div_x_t_kernel_x_t_phi}
    for op in OPS:
        got = views[op](X, dom, bdy)
        assert got.dtype == np.float16 and got.shape == (77, 4 * nd + nb)
        want = ogp._features(op, X.astype(np.float64))
        diff = np.abs(got.astype(np.float64) - want)
        # float64 arithmetic on both sides: the float16 rounding is decided alike except where exp() differs in its last bit at a midpoint
        assert (diff == 0).mean() > 0.999 and np.all(diff <= _f16_ulp(want)), (op, (diff != 0).mean(), diff.max())
    # at the collocation points themselves the rows ARE the Gram's rows (scasml_gp_gram_compat: the same per-pair arithmetic), bit for bit
    K = ogp.kernel_phi_phi(dom.astype(np.float64), bdy.astype(np.float64))
    rows = gp.kernel_x_t_phi(dom, dom, bdy).astype(np.float64)
    assert (np.abs(rows - K[:nd]) == 0).mean() > 0.999
    # a torch tensor in, a torch tensor out
    import torch
    t = gp.dt_x_t_kernel_x_t_phi(torch.from_numpy(X).cuda(), dom, bdy)
    assert isinstance(t, torch.Tensor) and t.dtype == torch.float16 and np.array_equal(t.cpu().numpy(), views["dt"](X, dom, bdy))
    # kernel_x_t_phi_single and predict: dot(row, right_vector).astype(float16) (models/GP.py:653-671)
    row = gp.kernel_x_t_phi_single(X[3])
    assert row.shape == (4 * nd + max(nb, 1),) and np.array_equal(row, gp.kernel_x_t_phi(X[3:4], gp.x_t_domain, gp.x_t_boundary)[0])
    u = float(row.astype(np.float64) @ gp.right_vector.reshape(-1))
    assert abs(float(gp.predict(X[3:4])[0, 0]) - u) <= 1.5 * _f16_ulp(np.float64(u))


@pytest.mark.parametrize("d,nd,nb", [(3, 20, 7), (20, 50, 11)])
def test_documented_operator_rows_match_the_closed_forms(d, nd, nb):
    from oracle.equation import GradDependentNonlinear, sample_points
    from oracle.gp import OracleGP
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    dom, bdy = sample_points(np.random.default_rng(5), d, nd, nb)
    dom, bdy = dom.astype(np.float32), bdy.astype(np.float32)
    gp = GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(d + 1), compat=None)
    ogp = OracleGP(GradDependentNonlinear(d + 1))
    ogp.x_t_domain, ogp.x_t_boundary = dom.astype(np.float64), bdy.astype(np.float64)
    ogp.N_domain, ogp.N_boundary = nd, nb
    X = _test_points(d, 41, seed=6)
    views = {"I": gp.kernel_x_t_phi, "lap": gp.laplacian_x_t_kernel_x_t_phi, "dt": gp.dt_x_t_kernel_x_t_phi, "div": gp.div_x_t_kernel_x_t_phi}
    for op in OPS:
        got, want = views[op](X, dom, bdy), ogp._features(op, X.astype(np.float64))
        assert got.dtype == np.float64 and np.all(np.abs(got - want) <= 1e-12 * (1.0 + np.abs(want))), op
    # the gradient rows contract to compute_gradient of the same right_vector (models/GP.py:673-687)
    rv = np.random.default_rng(7).normal(size=(4 * nd + nb, 1))
    ogp.right_vector = rv
    G = gp.dx_t_kernel_x_t_phi(X, dom, bdy)
    assert G.shape == (41, 4 * nd + nb, d + 1)
    assert np.allclose(np.einsum("imk,m->ik", G, rv[:, 0]), ogp.compute_gradient(X.astype(np.float64)), rtol=1e-10, atol=1e-10)
    # single pairs: every derivative kernel of the class surface is the oracle's block at that pair
    x, y = X[0], dom[2]
    for name, ox, oy in (("kappa", "I", "I"), ("dt_x_t_kappa", "dt", "I"), ("dt_y_t_kappa", "I", "dt"), ("div_x_kappa", "div", "I"), ("div_y_kappa", "I", "div"),
                         ("laplacian_x_t_kappa", "lap", "I"), ("laplacian_y_t_kappa", "I", "lap"), ("dt_x_t_dt_y_t_kappa", "dt", "dt"),
                         ("dt_x_t_div_y_kappa", "dt", "div"), ("dt_x_t_laplacian_y_t_kappa", "dt", "lap"), ("div_x_dt_y_t_kappa", "div", "dt"),
                         ("div_x_div_y_kappa", "div", "div"), ("div_x_laplacian_y_t_kappa", "div", "lap"), ("laplacian_x_t_dt_y_t_kappa", "lap", "dt"),
                         ("laplacian_x_t_div_y_kappa", "lap", "div"), ("laplacian_x_t_laplacian_y_t_kappa", "lap", "lap")):
        want = float(ogp.block(ox, oy, x[None].astype(np.float64), y[None].astype(np.float64))[0, 0])
        assert abs(float(getattr(gp, name)(x, y)) - want) <= 1e-12 * (1.0 + abs(want)), name
    g = gp.dx_t_kappa(x, y)
    a = ogp.a
    assert g.shape == (d + 1,) and np.allclose(g, -a * (x.astype(np.float64) - y) * float(gp.kappa(x, y)), rtol=1e-12, atol=1e-14)
    assert np.array_equal(gp.dy_t_kappa(x, y), -g)
    assert np.allclose(gp.kappa_kernel(X[:5], dom[:9]), ogp.block("I", "I", X[:5].astype(np.float64), dom[:9].astype(np.float64)), rtol=1e-12, atol=0)


def test_as_coded_gradient_rows_and_pair_kernels():
    d, idx, nd, nb = 20, [11, 17, 12, 6, 4], 40, 13
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=41)
    from oracle.equation import GradDependentNonlinear
    from oracle.gp_compat import OracleGPCompat
    plain = OracleGPCompat(GradDependentNonlinear(d + 1), idx, round16=False, round_factor=False, round_out=False)     # derivatives pass through the casts
    for k in ("x_t_domain", "x_t_boundary", "N_domain", "N_boundary", "phi_dim", "right_vector"):
        setattr(plain, k, getattr(ogp, k))
    X = _test_points(d, 29, seed=42)
    G = gp.dx_t_kernel_x_t_phi(X, gp.x_t_domain, gp.x_t_boundary)
    assert G.dtype == np.float16 and G.shape == (29, 4 * nd + nb, d + 1)
    rv = ogp.right_vector[:, 0]
    got = np.einsum("imk,m->ik", G.astype(np.float64), rv)
    want = plain.compute_gradient(X.astype(np.float64))
    # every entry of G is rounded once to float16 (:324): the contraction differs from the un-rounded gradient by <= 2^-11 of sum |c G|
    mag = np.einsum("imk,m->ik", np.abs(G.astype(np.float64)), np.abs(rv))
    assert np.all(np.abs(got - want) <= 2.0 ** -11 * mag + 1e-9)
    assert np.abs(got - want).max() > 0                                        # ... and the rounding is there
    x, y = X[1], gp.x_t_domain[4]
    for name, ox, oy in (("kappa", "I", "I"), ("laplacian_y_t_kappa", "I", "lap"), ("laplacian_x_t_kappa", "lap", "I"), ("dt_x_t_laplacian_y_t_kappa", "dt", "lap"),
                         ("div_x_laplacian_y_t_kappa", "div", "lap"), ("laplacian_x_t_div_y_kappa", "lap", "div"), ("laplacian_x_t_laplacian_y_t_kappa", "lap", "lap"),
                         ("div_x_div_y_kappa", "div", "div"), ("dt_x_t_dt_y_t_kappa", "dt", "dt")):
        want1 = float(ogp.block(ox, oy, x[None].astype(np.float64), y[None].astype(np.float64))[0, 0])
        got1 = getattr(gp, name)(x, y)
        assert got1.dtype == np.float16 and abs(float(got1) - want1) <= _f16_ulp(np.float64(want1)), name
    with pytest.raises(ValueError):
        gp.kernel_x_t_phi(X, np.zeros((0, d + 1), dtype=np.float32), gp.x_t_boundary)
