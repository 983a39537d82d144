"""The reference-compat modes on the HIP path against the oracle:

* GP(compat="reference", laplacian_idx=...)  -- shifted 5-index Hutchinson "Laplacian", float16 kernel entries, float16 K_p
  (csrc/gp_compat.hip vs oracle/gp_compat.py; models/GP.py:28-39, 43, 87-179, 266-268, 599, 719)
* compat_crn=True on the four solvers          -- the reference's key reuse as counter keying
  (csrc/picard_tree.hip SCASML_RNG_COMPAT_CRN vs oracle/mlp.py compat_crn; MLP.py:167-168,178, MLP_full_history.py:92-93,99,138)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDX20 = [11, 17, 12, 6, 4]


def _points(d, nd, nb, seed, f16=True):
    from oracle.equation import sample_points
    dom, bdy = sample_points(np.random.default_rng(seed), d, nd, nb)
    if f16:                                               # the reference's collocation arrays are float16 (experiment_run.py:30)
        dom, bdy = dom.astype(np.float16).astype(np.float32), bdy.astype(np.float16).astype(np.float32)
    return dom, bdy


def _pair(d, idx, nd, nb, seed):
    from oracle.equation import GradDependentNonlinear
    from oracle.gp_compat import OracleGPCompat
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    dom, bdy = _points(d, nd, nb, seed)
    eq = Grad_Dependent_Nonlinear(d + 1)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=idx)
    ogp = OracleGPCompat(GradDependentNonlinear(d + 1), idx, round16=True, round_factor=False)
    return gp, ogp, dom, bdy, eq


@pytest.mark.parametrize("d,idx,nd,nb", [(20, IDX20, 70, 25), (7, [5, 0, 3, 6, 2], 40, 9), (100, [99, 0, 50, 7, 31], 33, 7)])
def test_compat_gram_is_the_oracles_bit_for_bit_after_float16_rounding(d, idx, nd, nb):
    gp, ogp, dom, bdy, _ = _pair(d, idx, nd, nb, seed=0)
    K = gp.kernel_phi_phi(dom, bdy).cpu().numpy()
    Ko = ogp.kernel_phi_phi(dom, bdy)
    M = 4 * nd + nb
    off = ~np.eye(M, dtype=bool)
    # entries are float16 values: a float64 evaluation on either side rounds to the same one except within ~1e-13
    # (relative) of a rounding boundary
    assert np.mean(K[off] != Ko[off]) < 1e-4
    assert np.abs(K[off] - Ko[off]).max() <= 2.0 ** -10 * np.abs(Ko).max()
    from oracle.gp_compat import f16
    assert np.array_equal(np.diag(K), f16(np.diag(Ko) + ogp.nugget))          # float16(K_p) on the diagonal


@pytest.mark.parametrize("d,idx,nd,nb", [(20, IDX20, 120, 40), (7, [5, 0, 3, 6, 2], 30, 10)])
def test_compat_training_and_evaluation_match_oracle(d, idx, nd, nb):
    gp, ogp, dom, bdy, _ = _pair(d, idx, nd, nb, seed=2)
    gp.compat_eval = "float64"              # this test pins the float64 kernels (rounding decisions as NumPy takes them);
    ogp.round_out = False                   # the matrix-core kernel and the float16 outputs: tests/test_gpu_compat_mfma.py
    gp.GPsolver(dom, bdy, GN_steps=20)
    ogp.GPsolver(dom, bdy, GN_steps=20)
    assert len(gp.loss_history) == len(ogp.loss_history)
    assert np.allclose(gp.loss_history, ogp.loss_history, rtol=1e-7)
    rv, rvo = gp.right_vector, ogp.right_vector
    assert np.abs(rv - rvo).max() <= 1e-6 * np.abs(rvo).max()
    X = np.random.default_rng(4).uniform(-0.6, 0.6, (300, d + 1)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1])
    gp.load_right_vector(dom, bdy, ogp.right_vector)          # same coefficients on both sides from here on
    mag = (np.abs(ogp._features("I", X)) @ np.abs(ogp.right_vector))[:, 0] + 1e-3
    a = ogp.a
    magp = mag * (1 + a * (1 + d)) ** 2
    # float64 on both sides; a float16 rounding decision can differ for an entry within 1e-13 of a boundary,
    # which moves the sum by 2^-11 of ONE term -- tolerate a few such flips per point
    tol = 3 * 2.0 ** -11 * np.abs(ogp.right_vector).max()
    r16 = lambda v: 2.0 ** -11 * np.abs(v)             # the product returns u_hat and eps_PDE as float16 values (models/GP.py:671, 769)
    assert np.all(np.abs(gp.predict(X)[:, 0] - ogp.predict(X)[:, 0]) <= 1e-6 * mag + tol + r16(ogp.predict(X)[:, 0]))
    pts = gp._points_device(X)[0]
    out4 = gp._eval_device(pts).cpu().numpy()
    dt, div, lap = ogp.pde_parts(X)
    assert np.all(np.abs(out4[:, 1] - div[:, 0]) <= 1e-6 * magp + tol * a * d)
    assert np.all(np.abs(out4[:, 3] - dt[:, 0]) <= 1e-6 * magp + tol * a * d)
    eps = ogp.compute_PDE_loss(X)[:, 0]
    assert np.all(np.abs(out4[:, 2] - eps) <= 1e-6 * magp + tol * (a * d) ** 2 + r16(eps) + 2.0 ** -11 * np.abs(out4[:, 1]) * ogp.sigma_eq ** 2)


def test_compat_surrogate_differs_from_the_exact_one_and_needs_an_index_set():
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(21)
    default = GP_Grad_Dependent_Nonlinear(eq)                  # the reference's surrogate with the reference's own index draw
    assert default.compat == "reference" and default.laplacian_idx.tolist() == [0, 1, 19, 8, 12]
    with pytest.raises(ValueError):
        GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=None)
    with pytest.raises(ValueError):
        GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(4))       # d = 3 < 5 indices
    with pytest.raises(ValueError):
        GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=[1, 1, 2, 3, 4])
    with pytest.raises(ValueError):
        GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=[1, 20, 2, 3, 4])
    dom, bdy = _points(20, 200, 50, seed=5)
    exact = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    exact.GPsolver(dom, bdy)
    compat = GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=IDX20)
    compat.GPsolver(dom, bdy)
    X = np.concatenate(_points(20, 150, 50, seed=6))
    pe, pc = exact.predict(X), compat.predict(X)
    assert 1e-4 < np.abs(pe - pc).max() < 0.1


def test_scasml_on_the_compat_surrogate_matches_oracle():
    """Solver logic on the as-coded surrogate.  The surrogate's outputs are float16 VALUES (models/GP.py:671, 769), so one entry whose
    float16 rounding is decided differently moves u_hat by a float16 ulp (2.4e-4 .. 4.9e-4) and a z component by that times
    N / (MC delta_t): element-wise agreement with the float64 oracle is asked of the float64 evaluation kernel (decisions differ only
    within 1e-13 of a boundary); the matrix-core kernel (float32 values, ~1 decision in 500 differs) must agree with it on all but a
    few elements, and on those by no more than such a flip."""
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    gp, ogp, dom, bdy, eq = _pair(20, IDX20, 60, 20, seed=7)
    ogp.GPsolver(dom, bdy, GN_steps=20)
    gp.load_right_vector(dom, bdy, ogp.right_vector)
    assert gp._compat_model is not None
    xt = np.concatenate(_points(20, 48, 16, seed=30, f16=False))
    for crn in (False, True):
        for cls, oracle_args, call, ocall in ((ScaSML, "quad", lambda s: s.uz_solve(2, 2, xt), lambda o: o.uz_solve(2, 2, xt)),
                                              (ScaSML_full_history, "fh", lambda s: s.uz_solve(2, None, xt, 3), lambda o: o.uz_solve(2, 3, xt))):
            want = ocall(PicardOracle(ogp.eq, oracle_args, gp=ogp, seed=3, stream=0, compat_crn=crn))
            gp.compat_eval = "float64"
            got64 = call(cls(eq, gp, seed=3, compat_crn=crn))
            gp.compat_eval = "mfma"
            got = call(cls(eq, gp, seed=3, compat_crn=crn))
            bad64 = np.abs(got64 - want) > 5e-5 + 2e-4 * np.abs(want)
            assert bad64.mean() <= 0.01, (cls.__name__, crn, bad64.mean(), np.abs(got64 - want).max())
            assert np.abs(got64[:, 0] - want[:, 0]).max() <= 3e-4
            bad = np.abs(got - got64) > 5e-5 + 2e-4 * np.abs(got64)
            assert bad.mean() <= 0.05, (cls.__name__, crn, bad.mean())
            assert np.abs(got[:, 0] - got64[:, 0]).max() <= 6e-4, np.abs(got[:, 0] - got64[:, 0]).max()
            assert np.abs(got - got64).max() <= 2e-2


@pytest.mark.parametrize("variant,d,n,par,B", [("quad", 20, 2, 2, 100), ("quad", 20, 3, 3, 40), ("quad", 100, 3, 3, 6), ("quad", 6, 4, 4, 9),
                                              ("fh", 20, 2, 3, 100), ("fh", 20, 4, 3, 12), ("fh", 100, 3, 2, 8)])
def test_common_random_number_keying_matches_oracle(variant, d, n, par, B):
    from oracle.equation import GradDependentNonlinear
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    xt = np.concatenate(_points(d, B - B // 4, B // 4, seed=40, f16=False))
    eq = Grad_Dependent_Nonlinear(d + 1)
    ora = PicardOracle(GradDependentNonlinear(d + 1), variant, seed=11, stream=0, compat_crn=True)
    if variant == "quad":
        hip, plain = MLP(eq, seed=11, compat_crn=True), MLP(eq, seed=11)
        got, base, want = hip.uz_solve(n, par, xt), plain.uz_solve(n, par, xt), ora.uz_solve(n, par, xt)
    else:
        hip, plain = MLP_full_history(eq, seed=11, compat_crn=True), MLP_full_history(eq, seed=11)
        got, base, want = hip.uz_solve(n, None, xt, par), plain.uz_solve(n, None, xt, par), ora.uz_solve(n, par, xt)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.all(np.abs(got[m] - want[m]) <= 2e-5 + 1e-4 * np.abs(want[m])), np.abs(got[m] - want[m]).max()
    if variant == "quad":
        assert np.abs(got[m] - base[m]).max() > 1e-3      # and it IS a different estimator from the independent-draw default
    else:
        # plain full history: the only shared draws are the level-0 normals, whose term is f(0, 0) = 0 for this PDE
        # (the level-0 child is uz(0) = 0, MLP_full_history.py:119-121), so the keying cannot show; the ScaSML test above covers it
        assert np.array_equal(got[m], base[m])


def test_points_one_float16_ulp_below_terminal_time():
    """close to T the terminal normals cannot be recovered from the stored X_T accurately; ACCUMULATE replays
    them below kReadbackMinVol.  t = T - 2.4e-4 is the largest float16 below T = 0.5."""
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    gp, ogp, dom, bdy, eq = _pair(20, IDX20, 60, 20, seed=8)
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    ogp = OracleGP(GradDependentNonlinear(21))
    ogp.GPsolver(dom, bdy)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    gp.load_right_vector(dom, bdy, ogp.right_vector)
    xt = np.concatenate(_points(20, 48, 16, seed=31, f16=False))
    xt[:, -1] = np.float32(np.nextafter(np.float16(0.5), np.float16(0)))
    xt[::3, -1] = 0.4985                                     # T - t = 1.5e-3: just under the replay threshold
    xt[1::3, -1] = 0.498                                     # just over it
    hip = ScaSML(eq, gp, seed=5)
    got, want = hip.uz_solve(2, 2, xt), PicardOracle(ogp.eq, "quad", gp=ogp, seed=5, stream=0).uz_solve(2, 2, xt)
    assert np.all(np.abs(got - want) <= 5e-5 + 2e-4 * np.abs(want)), np.abs(got - want).max()


@pytest.mark.parametrize("variant,n,par", [("quad", 2, 2), ("quad", 3, 3), ("fh", 2, 3)])
def test_solver_level_float16_casts_match_oracle(variant, n, par):
    """compat_f16 (SCASML_RNG_COMPAT_F16): Equation.g / Equation.f return float16, ScaSML.g / ScaSML.f subtract float16 values, and every
    uz_solve returns clip(...).astype(float16) -- except ScaSML_full_history (solvers/ScaSML_full_history.py:199).  A cast decided on a
    float32 value here and a float64 value in the oracle can differ by one float16 ulp on rare elements."""
    from oracle.equation import GradDependentNonlinear
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    d = 20
    gp, ogp, dom, bdy, eq = _pair(d, IDX20, 60, 20, seed=7)
    ogp.GPsolver(dom, bdy, GN_steps=20)
    gp.load_right_vector(dom, bdy, ogp.right_vector)
    gp.compat_eval = "float64"                       # rounding decisions of the surrogate as the oracle takes them
    xt = np.concatenate(_points(d, 48, 16, seed=33, f16=False))
    is16 = lambda v: np.array_equal(v.astype(np.float16).astype(v.dtype), v)
    for scasml in (False, True):
        if variant == "quad":
            hip = ScaSML(eq, gp, seed=4, compat_f16=True) if scasml else MLP(eq, seed=4, compat_f16=True)
            plain = ScaSML(eq, gp, seed=4) if scasml else MLP(eq, seed=4)
            got, base = hip.uz_solve(n, par, xt), plain.uz_solve(n, par, xt)
        else:
            hip = ScaSML_full_history(eq, gp, seed=4, compat_f16=True) if scasml else MLP_full_history(eq, seed=4, compat_f16=True)
            plain = ScaSML_full_history(eq, gp, seed=4) if scasml else MLP_full_history(eq, seed=4)
            got, base = hip.uz_solve(n, None, xt, par), plain.uz_solve(n, None, xt, par)
        ora = PicardOracle(ogp.eq, variant, gp=ogp if scasml else None, seed=4, stream=0, compat_f16=True)
        want = ora.uz_solve(n, par, xt)
        assert is16(got) == (not (scasml and variant == "fh")), (variant, scasml)
        assert not np.array_equal(got, base)                         # the casts do something
        diff = np.abs(got - want)
        ulp = 2.0 ** -10 * np.maximum(np.abs(want), 2.0 ** -14)      # one float16 ulp
        assert (diff > ulp + 1e-6).mean() <= 0.03, (variant, scasml, (diff > ulp + 1e-6).mean())
        assert np.abs(got[:, 0] - want[:, 0]).max() <= 6e-4 and diff.max() <= 2e-2, (variant, scasml, diff.max())


def test_fit_at_the_references_own_size_matches_the_oracles_fit():
    """The training half at the size the reference runs and the bench measures (d = 100, 1000 + 200 collocation points, M = 4200, the harness's
    seeded training set): the device fit -- as-coded Gram, float64 Cholesky, Newton on the explicit inverse -- against the oracle's own fit of the
    same points (oracle/gp_compat.py: eigh factor, NumPy Newton; ~20 s of CPU).  Same number of Newton steps, the loss history to 1e-6, right_vector
    to 1e-9 of its largest entry (measured 8e-12), float16 predictions equal but for a few flips of one float16 ulp.  (The smaller cases above bound right_vector
    at 1e-6; the logged GP errors pin this size through predictions only.)"""
    from oracle.equation import GradDependentNonlinear
    from oracle.gp_compat import OracleGPCompat
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    d = 100
    eq = Grad_Dependent_Nonlinear(d + 1)
    np.random.seed(1234)
    dom, bdy = eq.generate_data(1000, 200)
    xt = np.concatenate(eq.generate_test_data(1000, 200))
    gp = GP_Grad_Dependent_Nonlinear(eq, compat="reference")
    gp.GPsolver(dom, bdy, GN_steps=20)
    ogp = OracleGPCompat(GradDependentNonlinear(d + 1), gp.laplacian_idx, round16=True, round_factor=False)
    ogp.GPsolver(np.asarray(dom, dtype=np.float64), np.asarray(bdy, dtype=np.float64), GN_steps=20)
    assert gp.phi_dim == 4200 and len(gp.loss_history) == len(ogp.loss_history)
    assert np.allclose(gp.loss_history, ogp.loss_history, rtol=1e-6)
    rv, rvo = np.asarray(gp.right_vector, dtype=np.float64), np.asarray(ogp.right_vector, dtype=np.float64)
    worst = float(np.abs(rv - rvo).max() / np.abs(rvo).max())
    pg, po = gp.predict(xt).astype(np.float64)[:, 0], ogp.predict(xt.astype(np.float64))[:, 0]
    flips = pg != po
    print("M = 4200 fit: right_vector max diff / max %.3e, %d Newton steps, %d of %d float16 predictions differ (max %.2e)"
          % (worst, len(gp.loss_history) - 1, int(flips.sum()), len(pg), float(np.abs(pg - po).max())))
    assert worst <= 1e-9                                                       # measured 8.2e-12
    assert flips.mean() <= 0.25 and np.abs(pg - po).max() <= 2.0 ** -10        # float16 values of |u| <= 1: a flip is at most one ulp (4.9e-4) ... two near 1
