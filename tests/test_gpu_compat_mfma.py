"""The reference's as-coded surrogate on the matrix cores (csrc/gp_eval_compat_mfma.hip) against the float64 statements:
oracle/gp_compat.py (NumPy) and csrc/gp_compat.hip (device float64).

* with the float16 rounding of the entries switched OFF both sides are smooth functions and must agree to float32 accuracy of
  the sums -- this pins the formulas of the three shifted geometries and the Hutchinson product;
* with it ON an entry's rounding decision is taken on a float32 value here and on a float64 value there: it can differ where the
  value lies within ~2^-20 (relative) of a float16 midpoint, which moves ONE term of a sum by one float16 ulp -- tolerated per point;
* what a site consumes (u_hat, or u_hat and div) does not depend on which form its workgroup ran: bitwise.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(20, [11, 17, 12, 6, 4], 150, 50), (7, [5, 0, 3, 6, 2], 40, 9), (100, [99, 0, 50, 7, 31], 90, 38), (250, [249, 1, 100, 17, 200], 40, 24),
         (5, [0, 1, 2, 3, 4], 33, 31)]


def _setup(d, idx, nd, nb, seed, round16=True):
    from oracle.equation import GradDependentNonlinear, sample_points
    from oracle.gp_compat import OracleGPCompat
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    dom, bdy = sample_points(np.random.default_rng(seed), d, nd, nb)
    dom, bdy = dom.astype(np.float16).astype(np.float32), bdy.astype(np.float16).astype(np.float32)   # experiment_run.py:30
    eq = Grad_Dependent_Nonlinear(d + 1)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx=idx)
    ogp = OracleGPCompat(GradDependentNonlinear(d + 1), idx, round16=round16, round_factor=False)
    # coefficients of realistic size without a fit: K_p^-1 z of a trained model has entries up to |z| / nugget
    rng = np.random.default_rng(seed + 1)
    M = 4 * nd + nb
    rv = rng.normal(size=M) * np.concatenate([np.full(nd, 1.0), np.full(nb, 1.0), np.full(nd, 0.05), np.full(nd, 0.3), np.full(nd, 0.3)])
    ogp.x_t_domain, ogp.x_t_boundary = dom.astype(np.float64), bdy.astype(np.float64)
    ogp.N_domain, ogp.N_boundary, ogp.phi_dim = nd, nb, M
    ogp.right_vector = rv[:, None]
    gp.load_right_vector(dom, bdy, rv)
    assert gp._compat_model is not None
    return gp, ogp, eq


def _test_points(d, n, seed, spread=0.6):
    X = np.random.default_rng(seed).uniform(-spread, spread, (n, d + 1)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1]) * 0.8
    return X


def _raw(gp, X, round16, kinds=None, rows_per_site=0):
    """(out4, lap) straight from scasml_gp_eval_compat_sites."""
    import torch
    from scasml_gp_amd import _lib
    lib = _lib.load()
    pts = gp._points_device(X)[0]
    out4 = torch.zeros((pts.shape[0], 4), dtype=torch.float32, device="cuda")
    lap = torch.zeros((pts.shape[0],), dtype=torch.float32, device="cuda")
    kd = torch.from_numpy(np.asarray(kinds, dtype=np.uint8)).cuda() if kinds is not None else None
    _lib.check(lib.scasml_gp_eval_compat_sites(
        gp.d, 1.0 / float(gp.sigma) ** 2, float(gp.equation.sigma()), float(gp.equation.mu()), int(gp.equation.eq_id), _lib.ptr(gp._compat_model),
        gp.N_domain, gp.N_boundary, gp.laplacian_idx.ctypes.data_as(C.c_void_p), round16, 0.0, _lib.ptr(pts), pts.shape[0], rows_per_site,
        _lib.ptr(kd) if kd is not None else None, _lib.ptr(out4), _lib.ptr(lap), _lib.stream_ptr()), "gp_eval_compat_sites")
    return out4.cpu().numpy().astype(np.float64), lap.cpu().numpy().astype(np.float64)


def _magnitudes(ogp, X):
    """sum_j |c_j P_j| per operator row: the cancellation-free size of each sum."""
    rv = np.abs(ogp.right_vector)
    return {op: (np.abs(ogp._features(op, X)) @ rv)[:, 0] + 1e-6 for op in ("I", "dt", "div", "lap")}


@pytest.mark.parametrize("d,idx,nd,nb", CASES)
def test_formulas_without_rounding_match_the_float64_statement(d, idx, nd, nb):
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=3, round16=False)
    X = _test_points(d, 333, seed=8, spread=1.3 if d <= 20 else 0.7)
    out4, lap = _raw(gp, X, round16=0)
    mag = _magnitudes(ogp, X)
    dt, div, lp = ogp.pde_parts(X)
    # float32 products and sums against float64: 2e-5 of the cancellation-free magnitude (the bound of the exact-operator kernel)
    assert np.all(np.abs(out4[:, 0] - ogp.predict(X)[:, 0]) <= 2e-5 * mag["I"])
    assert np.all(np.abs(out4[:, 3] - dt[:, 0]) <= 2e-5 * mag["dt"])
    assert np.all(np.abs(out4[:, 1] - div[:, 0]) <= 2e-5 * mag["div"])
    assert np.all(np.abs(lap - lp[:, 0]) <= 2e-5 * mag["lap"])
    eps = ogp.compute_PDE_loss(X)[:, 0]
    s2 = ogp.sigma_eq ** 2
    assert np.all(np.abs(out4[:, 2] - eps) <= 2e-5 * (mag["dt"] + (abs(ogp.eq.mu()) + s2) * mag["div"] + 0.5 * s2 * mag["lap"] + mag["I"]))


@pytest.mark.parametrize("d,idx,nd,nb", CASES[:4])
def test_rounded_entries_match_the_float64_statement_up_to_rounding_flips(d, idx, nd, nb):
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=5, round16=True)
    X = _test_points(d, 257, seed=9)
    out4, lap = _raw(gp, X, round16=1)                    # entries rounded, outputs not
    ogp.round_out = False
    mag = _magnitudes(ogp, X)
    dt, div, lp = ogp.pde_parts(X)
    # a flipped rounding decision moves one term by 2^-11 of itself: allow a few of the largest per point
    flip = {op: 4 * 2.0 ** -11 * (np.abs(ogp._features(op, X)) * np.abs(ogp.right_vector)[:, 0][None, :]).max(1) for op in mag}
    err_rounded = np.abs(out4[:, 0] - ogp.predict(X)[:, 0])
    assert np.all(err_rounded <= 2e-5 * mag["I"] + flip["I"])
    assert np.all(np.abs(out4[:, 3] - dt[:, 0]) <= 2e-5 * mag["dt"] + flip["dt"])
    assert np.all(np.abs(out4[:, 1] - div[:, 0]) <= 2e-5 * mag["div"] + flip["div"])
    assert np.all(np.abs(lap - lp[:, 0]) <= 2e-5 * mag["lap"] + flip["lap"])
    # the rounding is there at all: the unrounded statement is much further away than the rounded one
    ogp.round16 = False
    assert np.abs(out4[:, 0] - ogp.predict(X)[:, 0]).mean() > 3 * err_rounded.mean()
    # float16 outputs (bit 1): u_hat is a float16 value and eps_PDE is formed from it
    ogp.round16, ogp.round_out = True, True
    out4r, _ = _raw(gp, X, round16=3)
    assert np.array_equal(out4r[:, 0].astype(np.float16).astype(np.float64), out4r[:, 0])
    assert np.array_equal(out4r[:, 2].astype(np.float16).astype(np.float64), out4r[:, 2])
    assert np.all(np.abs(out4r[:, 0] - out4[:, 0]) <= 2.0 ** -11 * np.abs(out4[:, 0]) + 1e-7)
    # and the device float64 kernel agrees with both
    gp.compat_eval = "float64"
    pts = gp._points_device(X)[0]
    ref = gp._eval_device(pts).cpu().numpy().astype(np.float64)
    assert np.all(np.abs(ref[:, 0] - ogp.predict(X)[:, 0]) <= 1e-6 * mag["I"] + flip["I"] + 2.0 ** -11 * np.abs(ref[:, 0]))
    assert np.all(np.abs(ref[:, 1] - out4r[:, 1]) <= 2e-5 * mag["div"] + flip["div"])


def test_what_a_site_consumes_does_not_depend_on_the_form_its_workgroup_ran():
    d, idx, nd, nb = 20, [11, 17, 12, 6, 4], 100, 33
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=6)
    rows = 128                                           # one workgroup per site
    kinds = [0, 1, 3, 4, 0, 4, 1, 2, 3]
    X = _test_points(d, rows * len(kinds), seed=10)
    full, lap_full = _raw(gp, X, round16=3)
    part, _ = _raw(gp, X, round16=3, kinds=kinds, rows_per_site=rows)
    for s, k in enumerate(kinds):
        sl = slice(s * rows, (s + 1) * rows)
        if k == 2:                                       # another rank's site: untouched (zeros of the fresh buffer)
            assert not part[sl].any()
            continue
        assert np.array_equal(part[sl, 0], full[sl, 0]), k            # u_hat: every form
        if k in (0, 4):
            assert np.array_equal(part[sl, 1], full[sl, 1]), k        # div: full and (u, div) forms
        if k == 0:
            assert np.array_equal(part[sl], full[sl])
    # sites narrower than a workgroup: the workgroup runs the most demanding form of the sites it spans -- same values again
    part32, _ = _raw(gp, X[:32 * len(kinds)], round16=3, kinds=kinds, rows_per_site=32)
    full32, _ = _raw(gp, X[:32 * len(kinds)], round16=3)
    for s, k in enumerate(kinds):
        sl = slice(s * 32, (s + 1) * 32)
        if k != 2:
            assert np.array_equal(part32[sl, 0], full32[sl, 0])
        if k in (0, 4):
            assert np.array_equal(part32[sl, 1], full32[sl, 1])


@pytest.mark.parametrize("rows,round16", [(32, 3), (96, 3), (128, 3), (160, 6)])
def test_a_site_list_gives_every_listed_row_the_same_bits_in_any_order(rows, round16):
    """scasml_gp_eval_compat_site_list (ABI 7): the grid covers the listed sites only, in the list's order.  A listed row gets what
    scasml_gp_eval_compat_sites gives it -- bit for bit in the as-coded mode, whatever the order and whichever sites share its workgroup (96-
    and 160-row sites: workgroups straddle sites that are neighbours in the LIST, not in the buffer) -- rows of sites not listed stay untouched,
    and malformed lists are refused before a launch."""
    import torch
    from scasml_gp_amd import _lib
    d, idx, nd, nb = 20, [11, 17, 12, 6, 4], 100, 33
    gp, _, _ = _setup(d, idx, nd, nb, seed=6)
    kinds = [0, 1, 3, 4, 0, 4, 1, 2, 3, 0, 4]
    X = _test_points(d, rows * len(kinds), seed=10)
    base, _ = _raw(gp, X, round16=round16, kinds=kinds, rows_per_site=rows)
    lib = _lib.load()
    pts = gp._points_device(X)[0]
    kd = torch.from_numpy(np.asarray(kinds, dtype=np.uint8)).cuda()

    def run(order, n_inf=None, rps=rows, n_listed=None):
        out4 = torch.full((pts.shape[0], 4), -7.0, dtype=torch.float32, device="cuda")
        od = torch.from_numpy(np.asarray(order, dtype=np.int32)).cuda()
        rc = lib.scasml_gp_eval_compat_site_list(
            gp.d, 1.0 / float(gp.sigma) ** 2, float(gp.equation.sigma()), float(gp.equation.mu()), int(gp.equation.eq_id), _lib.ptr(gp._compat_model),
            gp.N_domain, gp.N_boundary, gp.laplacian_idx.ctypes.data_as(C.c_void_p), round16, 0.0, _lib.ptr(pts), pts.shape[0] if n_inf is None else n_inf, rps,
            _lib.ptr(kd), _lib.ptr(od), len(order) if n_listed is None else n_listed, _lib.ptr(out4), None, _lib.stream_ptr())
        return rc, out4.cpu().numpy().astype(np.float64)

    mine = [s for s, k in enumerate(kinds) if k != 2]
    by_cost = sorted(mine, key=lambda s: {0: 0, 4: 1, 3: 2, 1: 2}[kinds[s]])           # the engine's launch order
    for order in (mine, by_cost, mine[::-1], [9, 2], [5]):
        rc, got = run(order)
        assert rc == 0
        for s, k in enumerate(kinds):
            sl = slice(s * rows, (s + 1) * rows)
            if s not in order:
                assert (got[sl] == -7.0).all(), (order, s)                             # not listed: not written
                continue
            if round16 & 1:
                assert np.array_equal(got[sl, 0], base[sl, 0]), (order, s, k)          # u_hat: every form
                if k in (0, 4):
                    assert np.array_equal(got[sl, 1], base[sl, 1]), (order, s, k)
                if k == 0:
                    assert np.array_equal(got[sl], base[sl]), (order, s)
            else:                                                                      # geometry mode: the forms order their sums differently
                assert np.allclose(got[sl, 0], base[sl, 0], rtol=0, atol=2e-5), (order, s, k)
    assert run(mine, n_listed=0)[0] == 0 and (run(mine, n_listed=0)[1] == -7.0).all()      # an empty list: nothing launched
    assert run(mine, rps=rows + 1)[0] == -1 and run(mine, n_inf=pts.shape[0] - 16)[0] == -1 and run(mine, n_listed=len(kinds) + 1)[0] == -1


@pytest.mark.parametrize("d,idx,nd,nb", CASES)
def test_geometry_mode_factored_sums_in_every_site_form(d, idx, nd, nb):
    """compat="reference-geometry" (round16 bit 0 off): the factored epilogue, per site form.  Every form's outputs against the float64
    statement without entry rounding, and what a site consumes is the same number in whichever form its workgroup ran -- to float32
    accuracy of the sums (the forms order their additions differently), not bitwise as in the as-coded mode."""
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=21, round16=False)
    rows = 128
    kinds = [0, 1, 3, 4, 0, 4, 1]
    X = _test_points(d, rows * len(kinds), seed=22, spread=0.7)
    mag = _magnitudes(ogp, X)
    dt, div, lp = ogp.pde_parts(X)
    u = ogp.predict(X)[:, 0]
    for bits in (0, 4):                                   # two point planes / one
        # one plane: the point enters x.y rounded to float16: |delta Lambda| <= 1.45 a 2^-12 sum_k |x_k y_k| per pair, relative in kappa
        a = 1.0 / float(gp.sigma) ** 2
        ycol = np.concatenate([ogp.x_t_domain, ogp.x_t_boundary])
        loose = 0.0 if bits == 0 else 1.0 * a * 2.0 ** -12 * float((np.abs(X).astype(np.float64) @ np.abs(ycol).T).max())
        tol = 2e-5 + loose
        part, _ = _raw(gp, X, round16=bits, kinds=kinds, rows_per_site=rows)
        full, lap = _raw(gp, X, round16=bits)
        assert np.all(np.abs(full[:, 0] - u) <= tol * mag["I"])
        assert np.all(np.abs(full[:, 3] - dt[:, 0]) <= tol * mag["dt"])
        assert np.all(np.abs(full[:, 1] - div[:, 0]) <= tol * mag["div"])
        assert np.all(np.abs(lap - lp[:, 0]) <= tol * mag["lap"])
        for s_, k in enumerate(kinds):
            sl = slice(s_ * rows, (s_ + 1) * rows)
            assert np.all(np.abs(part[sl, 0] - u[sl]) <= tol * mag["I"][sl]), (bits, k)
            if k in (0, 4):
                assert np.all(np.abs(part[sl, 1] - div[sl, 0]) <= tol * mag["div"][sl]), (bits, k)
            if k == 0:
                assert np.all(np.abs(part[sl, 3] - dt[sl, 0]) <= tol * mag["dt"][sl])
    # one plane is an approximation of the stated size, not a different function: against two planes
    two, _ = _raw(gp, X, round16=0)
    one, _ = _raw(gp, X, round16=4)
    assert np.all(np.abs(one[:, 0] - two[:, 0]) <= loose * mag["I"] + 1e-6)
    assert np.abs(one[:, 0] - two[:, 0]).max() > 0.0
    # ragged batches: a row's result does not depend on how many rows follow it (the last workgroup and wave are partly shadow rows)
    for n_rows in (1, 33, 333):
        part, lap_part = _raw(gp, X[:n_rows], round16=6)
        whole, lap_whole = _raw(gp, X, round16=6)
        assert np.array_equal(part, whole[:n_rows]) and np.array_equal(lap_part, lap_whole[:n_rows]), n_rows
    # the as-coded form refuses the one-plane option
    from scasml_gp_amd import _lib
    with pytest.raises(_lib.ScasmlError):
        _raw(gp, X[:32], round16=5)


def test_reference_geometry_surrogate_follows_the_reference_surrogate():
    """GP(compat="reference-geometry"): the fit of compat="reference" bit for bit; predictions and PDE residual within the rounding noise the
    as-coded evaluation carries (a float16 ulp of single terms), and a ScaSML solve on it within 1e-4 in relative L2 of the as-coded one."""
    from oracle.equation import sample_points
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    d = 40
    eq = Grad_Dependent_Nonlinear(d + 1)
    state = np.random.get_state()
    np.random.seed(7)
    dom, bdy = eq.generate_data(400, 80)
    xt = np.concatenate(eq.generate_test_data(500, 100))
    np.random.set_state(state)
    ref = GP_Grad_Dependent_Nonlinear(eq)
    geo = GP_Grad_Dependent_Nonlinear(eq, compat="reference-geometry")
    assert geo.compat == "reference" and geo.eval_geometry and geo.eval_round16 == 6 and ref.eval_round16 == 3
    ref.GPsolver(dom, bdy)
    geo.GPsolver(dom, bdy)
    assert np.array_equal(ref.right_vector, geo.right_vector)
    exact = np.asarray(eq.exact_solution(xt), dtype=np.float64)
    rel = lambda v: float(np.linalg.norm(np.asarray(v, dtype=np.float64) - exact) / np.linalg.norm(exact))
    pr, pg = ref.predict(xt).astype(np.float64), geo.predict(xt).astype(np.float64)
    assert np.abs(pr - pg).max() <= 8 * 2.0 ** -11 and abs(rel(pr) - rel(pg)) <= 1e-4
    er, eg = ref.compute_PDE_loss(xt).astype(np.float64), geo.compute_PDE_loss(xt).astype(np.float64)
    assert np.abs(er - eg).max() <= 0.02 * np.abs(er).max()
    sr = ScaSML(eq, ref, seed=3).u_solve(2, 2, xt)
    sg = ScaSML(eq, geo, seed=3).u_solve(2, 2, xt)
    assert abs(rel(sr) - rel(sg)) <= 1e-4, (rel(sr), rel(sg))
    # a state saved from one loads into the other: the fit is the same object
    geo2 = GP_Grad_Dependent_Nonlinear(eq, compat="reference-geometry").load_state_dict(ref.state_dict())
    assert np.array_equal(geo2.predict(xt), geo.predict(xt))


@pytest.mark.parametrize("d,idx,nd,nb", [(20, [11, 17, 12, 6, 4], 80, 20), (7, [5, 0, 3, 6, 2], 40, 9)])
def test_compat_gradient_is_the_gradient_of_the_as_coded_surrogate(d, idx, nd, nb):
    gp, ogp, _ = _setup(d, idx, nd, nb, seed=11)
    X = _test_points(d, 64, seed=12)
    got = gp.compute_gradient(X).astype(np.float64)
    want = ogp.compute_gradient(X)
    assert got.shape == (64, d + 1)
    assert np.array_equal(got, got.astype(np.float16).astype(np.float64))                    # .astype(float16), models/GP.py:687
    assert np.all(np.abs(got - want) <= 2.0 ** -10 * np.abs(want) + 1e-6)
    # its spatial columns sum to the div row of the fused evaluation up to the entries' float16 rounding
    out4, _ = _raw(gp, X, round16=1)
    mag = _magnitudes(ogp, X)
    assert np.all(np.abs(got[:, :d].sum(1) - out4[:, 1]) <= 2.0 ** -9 * mag["div"])
    assert np.all(np.abs(got[:, d] - out4[:, 3]) <= 2.0 ** -9 * mag["dt"])


def test_refusals():
    import torch
    from scasml_gp_amd import _lib
    from oracle.equation import sample_points
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    lib = _lib.load()
    gp, _, _ = _setup(20, [11, 17, 12, 6, 4], 40, 24, seed=13)
    pts = torch.zeros((8, 32), dtype=torch.float32, device="cuda")
    out = torch.zeros((8, 4), dtype=torch.float32, device="cuda")
    idx = gp.laplacian_idx.ctypes.data_as(C.c_void_p)
    bad = np.asarray([1, 1, 2, 3, 4], dtype=np.int32)
    args = lambda ix, xb: (20, 0.8, 0.25, -0.08, 0, _lib.ptr(gp._compat_model), 40, 24, ix, 3, xb, _lib.ptr(pts), 8, 0, None, _lib.ptr(out), None, None)
    assert lib.scasml_gp_eval_compat_sites(*args(bad.ctypes.data_as(C.c_void_p), 0.0)) == -1 and b"repeated" in lib.scasml_last_error()
    assert lib.scasml_gp_eval_compat_sites(*args(idx, 1e3)) == -2 and b"fp16 range" in lib.scasml_last_error()
    assert lib.scasml_gp_eval_compat_sites(*args(idx, 0.0)) == 0
    # collocation points that are not float16 values: no matrix-core model, the float64 kernel serves the call
    dom, bdy = sample_points(np.random.default_rng(0), 20, 40, 24)
    eq = Grad_Dependent_Nonlinear(21)
    g2 = GP_Grad_Dependent_Nonlinear(eq, compat="reference", laplacian_idx="original")
    assert g2.laplacian_idx.tolist() == [18, 17, 15, 9, 11]
    g2.load_right_vector(dom, bdy, np.ones(4 * 40 + 24))
    assert g2._compat_model is None and np.isfinite(g2.predict(dom)).all()
