"""BASELINE.json configs[2] at full size (d = 100, ScaSML n = rho = 3, 16384 roots, GP on 1000 + 200 points) through
size-independent properties, on both surrogates (the reference's as-coded one -- the default and the benchmarked path -- and the
documented operators); the oracle is consulted on a 64-root sample."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, N, B = 100, 3, 1 << 14


@pytest.fixture(scope="module", params=["reference", None], ids=["as-coded", "documented"])
def headline(request):
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    eq = Grad_Dependent_Nonlinear(D + 1)
    eq.geometry()
    state = np.random.get_state()
    np.random.seed(1234)
    x_dom, x_bdy = eq.generate_data(1000, 200)
    np.random.set_state(state)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=request.param)
    gp.GPsolver(x_dom, x_bdy, GN_steps=20)
    assert request.param is None or gp._compat_model is not None        # the matrix-core kernel serves the as-coded surrogate
    solver = ScaSML(eq, gp, seed=0)
    g = np.random.default_rng(1234)
    x_t = np.concatenate([g.uniform(-0.5, 0.5, (B, D)), g.uniform(0.0, 0.5, (B, 1))], axis=1).astype(np.float32)
    x_dev = torch.from_numpy(x_t).cuda()
    full, uhat, _ = solver._engine.solve(N, N, x_dev, stream_id=7)
    return solver, gp, x_dom, x_bdy, x_t, x_dev, full, uhat


def test_full_batch_is_finite_clipped_and_deterministic(headline):
    import torch
    solver, _, _, _, _, x_dev, full, uhat = headline
    assert full.shape == (B, D + 1) and bool(torch.isfinite(full).all()) and bool(torch.isfinite(uhat).all())
    clip = float(solver.equation.uncertainty)                             # ScaSML clips the defect at +-uncertainty (ScaSML.py:282-284)
    assert float(full.abs().max()) <= clip * (1 + 1e-6)
    again, uhat2, _ = solver._engine.solve(N, N, x_dev, stream_id=7)
    assert torch.equal(again, full) and torch.equal(uhat2, uhat)
    other, _, _ = solver._engine.solve(N, N, x_dev, stream_id=8)          # another call draws other normals
    assert not torch.equal(other, full)


def test_any_slice_of_roots_reproduces_its_rows_bitwise(headline):
    """Philox is keyed by the global root index and no reduction crosses roots: a slice of the batch solved on
    its own (root0 = its offset) must give the rows of the full solve bit for bit -- the property roots sharding
    across GPUs rests on."""
    import torch
    solver, _, _, _, _, x_dev, full, uhat = headline
    for lo, hi in [(0, 64), (5000, 5033), (B - 100, B)]:
        part, uh, _ = solver._engine.solve(N, N, x_dev[lo:hi], root0=lo, stream_id=7)
        assert torch.equal(part, full[lo:hi]) and torch.equal(uh, uhat[lo:hi])


def test_sample_sharded_partials_add_up_at_full_size(headline):
    import torch
    solver, _, _, _, _, x_dev, full, _ = headline
    eng = solver._engine
    total = None
    for r in range(2):
        part, _, _ = eng.solve(N, N, x_dev, rank=r, world=2, stream_id=7)
        total = part.clone() if total is None else total + part
    assert torch.allclose(eng.finalize_partials(total), full, atol=2e-5, rtol=1e-5)


def test_eight_ranks_share_the_samples_of_the_headline_shape(headline):
    """The north-star split at the headline shape (n = rho = 3): the root call's 77 units -- terminal samples and the "+" / "-" addends of the NODES of its
    sample paths -- dealt over EIGHT ranks by cost balance to 1 % (whole paths as units: 3.19), and the eight partial estimators add up to the unsharded
    one (one all-reduce in the multi-GPU run; here the ranks are walked in turn on 2048 of the roots)."""
    import torch
    solver, _, _, _, _, x_dev, full, _ = headline
    eng = solver._engine
    owner, _, load = eng.unit_owners(N, N, 8)
    assert len(owner) == 77 and set(owner.tolist()) == set(range(8)) and load.max() / load.mean() < 1.01
    sub = x_dev[4096:6144]
    total = None
    for r in range(8):
        part, _, _ = eng.solve(N, N, sub, root0=4096, rank=r, world=8, stream_id=7)
        total = part.clone() if total is None else total + part
    assert torch.allclose(eng.finalize_partials(total), full[4096:6144], atol=2e-5, rtol=1e-5)


def test_sample_of_the_full_batch_matches_the_oracle(headline):
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.gp_compat import OracleGPCompat
    from oracle.mlp import PicardOracle
    solver, gp, x_dom, x_bdy, x_t, _, full, _ = headline
    oeq = GradDependentNonlinear(D + 1)
    ogp = OracleGPCompat(oeq, gp.laplacian_idx, round_factor=False) if gp.compat == "reference" else OracleGP(oeq)   # the same trained surrogate
    ogp.x_t_domain, ogp.x_t_boundary = np.asarray(x_dom, dtype=np.float64), np.asarray(x_bdy, dtype=np.float64)
    ogp.N_domain, ogp.N_boundary = len(x_dom), len(x_bdy)
    ogp.phi_dim = 4 * len(x_dom) + len(x_bdy)
    ogp.right_vector = gp.right_vector
    ora = PicardOracle(oeq, "quad", gp=ogp, seed=0, stream=7)
    got, want = [], []
    for lo in (0, 4090, 8184, B - 16):                     # 4 x 16 roots spread over the batch
        want.append(ora.uz_solve(N, N, x_t[lo:lo + 16], root0=lo))
        got.append(full[lo:lo + 16].cpu().numpy())
        if gp.compat == "reference":
            # every element accounted for (tests/_explained_parity.py): oracle == the float64-kernel device run, tightly; the matrix-core run
            # differs from it only through u_hat / eps_PDE values one float16 ulp apart; and the slice reproduces the full batch's rows
            from _explained_parity import assert_explained
            again = assert_explained(solver._engine, N, N, x_t[lo:lo + 16], lo, 7, want[-1])
            assert np.array_equal(again, got[-1].astype(np.float64))
    got, want = np.concatenate(got), np.concatenate(want)
    diff = np.abs(got - want)
    if gp.compat == "reference":
        assert diff[:, 0].max() < 6e-4
        assert abs(np.linalg.norm(got[:, 0]) - np.linalg.norm(want[:, 0])) <= 1e-3 * np.linalg.norm(want[:, 0])
    else:
        assert diff.max() < 1e-4                           # outputs are clipped to +-0.1: 1e-3 relative to the clip


def test_config1_mlp_d20_n2_at_its_full_batch_of_2_20_roots():
    """BASELINE.json configs[1] at the batch bench.py's other_runs times (d = 20, solvers.MLP n = rho = 2, 2^20 roots = 4.8e7 path-steps by the reference's count): the
    size-independent properties -- finite, clipped at norm_estimation, deterministic, another call draws other normals, any slice solved on
    its own reproduces its rows bit for bit -- and the oracle on 64 roots spread over the batch."""
    import torch
    from oracle.equation import GradDependentNonlinear
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    d, n, big = 20, 2, 1 << 20
    eq = Grad_Dependent_Nonlinear(d + 1)
    eq.geometry()
    g = np.random.default_rng(1234)
    x_t = np.concatenate([g.uniform(-0.5, 0.5, (big, d)), g.uniform(0.0, 0.5, (big, 1))], axis=1).astype(np.float32)
    x_dev = torch.from_numpy(x_t).cuda()
    eng = MLP(eq, seed=0)._engine
    full, none, _ = eng.solve(n, n, x_dev, stream_id=3)
    assert none is None and full.shape == (big, d + 1) and bool(torch.isfinite(full).all())
    assert float(full.abs().max()) <= float(eq.norm_estimation) * (1 + 1e-6)          # MLP.py:272-274
    again, _, _ = eng.solve(n, n, x_dev, stream_id=3)
    other, _, _ = eng.solve(n, n, x_dev, stream_id=4)
    assert torch.equal(again, full) and not torch.equal(other, full)
    for lo, hi in [(0, 64), (500000, 500257), (big - 1000, big)]:                     # root sharding: a slice is its rows of the whole
        part, _, _ = eng.solve(n, n, x_dev[lo:hi], root0=lo, stream_id=3)
        assert torch.equal(part, full[lo:hi])
    ora = PicardOracle(GradDependentNonlinear(d + 1), "quad", seed=0, stream=3)
    for lo in (0, 349520, 699050, big - 16):
        want = ora.uz_solve(n, n, x_t[lo:lo + 16], root0=lo)
        got = full[lo:lo + 16].cpu().numpy()
        assert np.all(np.abs(got - want) <= 2e-5 + 1e-4 * np.abs(want)), np.abs(got - want).max()
    from scasml_gp_amd import tables
    # 46 path-steps per root as the reference counts them; 28 are executed (its n == 0 terminal draws are discarded work)
    assert tables.executed_path_steps(eng.plan(n, n)) == 28 and tables.reference_path_steps("quad", n, n) == 46
