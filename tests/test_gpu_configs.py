"""The BASELINE.json configurations that had no GPU test at their workload in round 1:

configs[0]  d = 10, GP only, 213 + 43 collocation points (M = 895), predict on 1000 + 200 points
configs[3]  d = 100, ScaSML_full_history n = 4, M = 3: oracle parity on 3 roots, and the full batch (2^14 roots)
            through size-independent properties incl. the 8-way Monte-Carlo sample split
configs[4]  staged: d = 250, GP on 1667 + 333 collocation points (M = 7001) + ScaSML n = rho = 3 against the oracle on 2
            roots; and a fit at 8333 + 1667 points (M = 34 999, K = 9.8 GB float64) checked through ||L L^T - K_p|| on sampled
            rows, loss descent and stationarity (the 1e5-point target needs the matrix sharded over 8 GPUs: scasml_gp_amd/dist_gp.py)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


SURROGATES = pytest.mark.parametrize("compat", ["reference", None], ids=["as-coded", "documented"])


def _oracle_with(gp, d):
    """The oracle surrogate carrying the coefficients the HIP fit produced (no second fit on the CPU): the reference's as-coded
    surrogate (oracle/gp_compat.py, same Hutchinson indices) for the default GP, the documented operators for GP(compat=None)."""
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.gp_compat import OracleGPCompat
    oeq = GradDependentNonlinear(d + 1)
    ogp = OracleGPCompat(oeq, gp.laplacian_idx, round_factor=False) if gp.compat == "reference" else OracleGP(oeq)
    ogp.x_t_domain = np.asarray(gp.x_t_domain, dtype=np.float64)
    ogp.x_t_boundary = np.asarray(gp.x_t_boundary, dtype=np.float64)
    ogp.N_domain, ogp.N_boundary = len(ogp.x_t_domain), len(ogp.x_t_boundary)
    ogp.phi_dim = 4 * ogp.N_domain + ogp.N_boundary
    ogp.right_vector = gp.right_vector
    return oeq, ogp


def _assert_close_to_oracle(eng, n, par, x_rows, root0, stream_id, got, want):
    """As-coded surrogate on the matrix cores: u_hat and eps_PDE are float16 VALUES, so a plain tolerance would have to be ~1e-2 wide on z.
    tests/_explained_parity.py accounts for every element instead (oracle == float64-kernel device run to 5e-5 + 2e-4 |want|; the product run
    differs from that one ONLY through u_hat / eps_PDE values that land one float16 ulp apart; negative controls).  Documented operators:
    the plain tight tolerance."""
    gp = eng.gp
    diff = np.abs(got - want)
    if gp.compat == "reference":
        from _explained_parity import assert_explained
        again = assert_explained(eng, n, par, x_rows, root0, stream_id, want)
        assert np.array_equal(again, got.astype(np.float64))      # the accounted-for run IS the run under test
        assert diff[:, 0].max() < 6e-4, diff[:, 0].max()          # u: at most a float16 ulp of u_hat's scale through the mean
    else:
        assert np.all(diff <= 5e-5 + 2e-4 * np.abs(want)), diff.max()


def test_config0_gp_only_d10_256_collocation_points():
    from oracle.equation import GradDependentNonlinear, rel_l2
    from oracle.gp import OracleGP
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    d = 10
    eq = Grad_Dependent_Nonlinear(d + 1)
    np.random.seed(1234)
    dom, bdy = eq.generate_data(213, 43)                         # 5 : 1 as tests/SimpleUniform.py:73-74
    xt = np.concatenate(eq.generate_test_data(1000, 200))
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    on_dom = gp.GPsolver(dom, bdy, GN_steps=20)
    ogp = OracleGP(GradDependentNonlinear(d + 1))
    want_dom = ogp.GPsolver(dom, bdy, GN_steps=20)
    assert gp.phi_dim == 895 and len(gp.loss_history) == len(ogp.loss_history)
    assert np.allclose(gp.loss_history, ogp.loss_history, rtol=1e-8)
    assert np.abs(gp.right_vector - ogp.right_vector).max() <= 1e-7 * np.abs(ogp.right_vector).max()
    assert np.allclose(on_dom, want_dom, atol=2e-5)
    got, want = gp.predict(xt), ogp.predict(xt)
    assert got.shape == (1200, 1) and np.abs(got - want).max() < 2e-5
    exact = eq.exact_solution(xt)
    assert abs(rel_l2(got, exact) - rel_l2(want, exact)) < 1e-5 and rel_l2(got, exact) < 0.2


# ---------------------------------------------------------------------------------------------- configs[3]
D3, N3, M3, B3 = 100, 4, 3, 1 << 14


@pytest.fixture(scope="module", params=["reference", None], ids=["as-coded", "documented"])
def config3(request):
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    eq = Grad_Dependent_Nonlinear(D3 + 1)
    state = np.random.get_state()
    np.random.seed(1234)
    dom, bdy = eq.generate_data(1000, 200)
    np.random.set_state(state)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=request.param)
    gp.GPsolver(dom, bdy, GN_steps=20)
    solver = ScaSML_full_history(eq, gp, seed=0)
    g = np.random.default_rng(4321)
    x_t = np.concatenate([g.uniform(-0.5, 0.5, (B3, D3)), g.uniform(0.0, 0.5, (B3, 1))], axis=1).astype(np.float32)
    x_dev = torch.from_numpy(x_t).cuda()
    full, uhat, _ = solver._engine.solve(N3, M3, x_dev, stream_id=5)
    return solver, gp, x_t, x_dev, full, uhat


def test_config3_full_history_n4_matches_oracle_on_sampled_roots(config3):
    from oracle.mlp import PicardOracle
    solver, gp, x_t, _, full, _ = config3
    oeq, ogp = _oracle_with(gp, D3)
    rows = np.array([0, 8191, B3 - 1])
    ora = PicardOracle(oeq, "fh", gp=ogp, seed=0, stream=5)
    want = np.concatenate([ora.uz_solve(N3, M3, x_t[r:r + 1], root0=int(r)) for r in rows])
    got = full[rows].cpu().numpy()
    if gp.compat == "reference":
        for r, g_row, w_row in zip(rows, got, want):
            _assert_close_to_oracle(solver._engine, N3, M3, x_t[r:r + 1], int(r), 5, g_row[None], w_row[None])
    else:
        assert np.max(np.abs(got - want)) < 1e-4          # outputs are clipped to +-0.1
    from scasml_gp_amd import tables
    assert tables.executed_path_steps(solver._engine.plan(N3, M3)) == ora.sites_executed == 1650
    assert tables.reference_path_steps("fh", N3, M3) == 2523


def test_config3_full_batch_properties_and_eight_way_sample_split(config3):
    import torch
    solver, _, _, x_dev, full, uhat = config3
    eng = solver._engine
    assert full.shape == (B3, D3 + 1) and bool(torch.isfinite(full).all()) and float(full.abs().max()) <= 0.1 * (1 + 1e-6)
    # a slice of roots solved on its own reproduces its rows bit for bit (root sharding)
    part, uh, _ = eng.solve(N3, M3, x_dev[7000:7040], root0=7000, stream_id=5)
    assert torch.equal(part, full[7000:7040]) and torch.equal(uh, uhat[7000:7040])
    # the north-star split: 240 Monte-Carlo units (terminal samples and the +/- addends of the samples) dealt over 8 ranks by cost, partial estimators add up to the full result
    sub = x_dev[:2048]
    total = None
    for r in range(8):
        p, _, _ = eng.solve(N3, M3, sub, rank=r, world=8, stream_id=5)
        total = p.clone() if total is None else total + p
    assert torch.allclose(eng.finalize_partials(total), full[:2048], atol=2e-5, rtol=1e-5)


# ---------------------------------------------------------------------------------------------- configs[4], staged
@SURROGATES
def test_config4_d250_two_thousand_collocation_points_scasml_n3_matches_oracle(compat):
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    d = 250
    eq = Grad_Dependent_Nonlinear(d + 1)
    np.random.seed(1234)
    dom, bdy = eq.generate_data(1667, 333)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
    gp.GPsolver(dom, bdy, GN_steps=20)
    assert gp.phi_dim == 7001 and gp.loss_history[-1] < gp.loss_history[0] and gp.grad_norms[-1] < 1e-3 * gp.grad_norms[0]
    assert compat is None or gp._compat_model is not None          # d = 250: 16 K-steps, the widest instantiation of the matrix-core kernel
    oeq, ogp = _oracle_with(gp, d)
    xt = np.concatenate(eq.generate_test_data(1, 1)).astype(np.float32)
    hip = ScaSML(eq, gp, seed=2)
    got = hip.uz_solve(3, 3, xt)
    want = PicardOracle(oeq, "quad", gp=ogp, seed=2, stream=0).uz_solve(3, 3, xt)
    _assert_close_to_oracle(hip._engine, 3, 3, xt, 0, 0, got, want)


def test_config4_fit_at_ten_thousand_collocation_points():
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    d = 250
    eq = Grad_Dependent_Nonlinear(d + 1)
    np.random.seed(1234)
    dom, bdy = eq.generate_data(8333, 1667)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    Kp = gp.kernel_phi_phi(dom, bdy)                      # K + nugget I (float64, 9.8 GB) and its factor
    M = gp.phi_dim
    assert M == 34999
    L = gp.cholesky_phi_phi_perturb
    rows = torch.from_numpy(np.random.default_rng(0).choice(M, 256, replace=False)).cuda()
    recon = L[rows] @ L.T                                  # 256 sampled rows of L L^T
    err = float((recon - Kp[rows]).abs().max())
    assert err <= 1e-11 * float(Kp.abs().max()) * 64, err
    del Kp, recon
    torch.cuda.empty_cache()
    gp.GPsolver(dom, bdy, GN_steps=20)
    h, g = gp.loss_history, gp.grad_norms
    assert h[-1] < h[0] and g[-1] < 1e-3 * g[0], (h, g)
    xt = np.concatenate(eq.generate_test_data(500, 100))
    from oracle.equation import rel_l2
    assert rel_l2(gp.predict(xt), eq.exact_solution(xt)) < 0.45
