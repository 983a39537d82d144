"""ScaSML (quadrature and full history) HIP pipeline -- generate points, fused GP evaluation,
accumulate -- against the oracle, which follows solvers/ScaSML.py literally (GP.predict +
full GP.compute_gradient per f call).  Outputs are clipped to +-0.1; GP values enter in
float32, so |diff| <= 5e-5 + 2e-4*|value|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL, RTOL = 5e-5, 2e-4


def _setup(d, nd, nb, variant, seed):
    from oracle.equation import GradDependentNonlinear, sample_points
    from oracle.gp import OracleGP
    from oracle.mlp import PicardOracle
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    dom, bdy = sample_points(np.random.default_rng(seed), d, nd, nb)
    oeq = GradDependentNonlinear(d + 1)
    ogp = OracleGP(oeq)
    ogp.GPsolver(dom, bdy, GN_steps=20)
    eq = Grad_Dependent_Nonlinear(d + 1)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    gp.GPsolver(dom, bdy, GN_steps=20)
    hip = ScaSML(eq, gp, seed=seed) if variant == "quad" else ScaSML_full_history(eq, gp, seed=seed)
    return hip, PicardOracle(oeq, variant, gp=ogp, seed=seed, stream=0), oeq


def _test_points(d, B, seed):
    from oracle.equation import sample_points
    return np.concatenate(sample_points(np.random.default_rng(seed), d, B - B // 4, B // 4))


@pytest.mark.parametrize("d,n,rho,B", [(10, 1, 1, 20), (20, 2, 2, 130), (20, 3, 3, 12), (100, 3, 3, 4), (6, 2, 3, 33)])
def test_scasml_quadrature_matches_oracle(d, n, rho, B):
    hip, ora, _ = _setup(d, 60, 20, "quad", seed=7)
    xt = _test_points(d, B, 30)
    got, want = hip.uz_solve(n, rho, xt), ora.uz_solve(n, rho, xt)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.all(np.abs(got[m] - want[m]) <= ATOL + RTOL * np.abs(want[m])), np.abs(got[m] - want[m]).max()


@pytest.mark.parametrize("d,n,M,B", [(20, 2, 3, 65), (20, 3, 2, 10), (100, 2, 3, 6)])
def test_scasml_full_history_matches_oracle(d, n, M, B):
    hip, ora, _ = _setup(d, 60, 20, "fh", seed=9)
    xt = _test_points(d, B, 31)
    got, want = hip.uz_solve(n, None, xt, M), ora.uz_solve(n, M, xt)
    assert np.all(np.abs(got - want) <= ATOL + RTOL * np.abs(want)), np.abs(got - want).max()


def test_u_solve_adds_surrogate_and_improves_on_it():
    from oracle.equation import rel_l2
    hip, ora, oeq = _setup(20, 300, 60, "quad", seed=5)
    xt = _test_points(20, 400, 32)
    u = hip.u_solve(2, 2, xt)
    want = ora.u_solve(2, 2, xt)
    assert u.shape == (400, 1) and np.allclose(u, want, atol=2e-4)
    exact = oeq.exact_solution(xt)
    assert rel_l2(u, exact) < rel_l2(hip.GP.predict(xt), exact)
    # level 0: defect is zero, u_solve returns the surrogate (ScaSML.py:217-219, 300-304)
    assert np.allclose(hip.u_solve(0, 2, xt), hip.GP.predict(xt), atol=1e-6)


def test_chunked_batches_equal_one_shot(monkeypatch):
    import scasml_gp_amd.solvers._picard as P
    hip, _, _ = _setup(20, 60, 20, "quad", seed=3)
    xt = _test_points(20, 50, 33)
    hip._engine.calls = 0
    one = hip.uz_solve(2, 2, xt)
    monkeypatch.setattr(P, "POINT_BUFFER_BYTES", 29 * 32 * 4 * 7)      # 7 roots per chunk
    hip._engine.calls = 0
    assert np.array_equal(one, hip.uz_solve(2, 2, xt))


def test_untrained_gp_fails_loudly():
    from scasml_gp_amd import _lib
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    eq = Grad_Dependent_Nonlinear(11)
    with pytest.raises(_lib.ScasmlError):
        ScaSML(eq, GP_Grad_Dependent_Nonlinear(eq, compat=None)).u_solve(1, 1, np.zeros((2, 11), dtype=np.float32))


def test_terminal_time_rows_float16_inputs_and_deepcopy():
    """t = T rows (tau = 0: every step degenerates, z saturates at the clip), float16 input arrays as the
    reference harness passes (experiment_run.py:30), and a deep-copied solver (tests/ComputingBudget.py:138)."""
    import copy
    hip, ora, _ = _setup(20, 60, 20, "quad", seed=4)
    xt = _test_points(20, 40, 34)
    xt[::5, -1] = 0.5
    got, want = hip.uz_solve(2, 2, xt), ora.uz_solve(2, 2, xt)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.all(np.abs(got[m] - want[m]) <= ATOL + RTOL * np.abs(want[m])), np.abs(got[m] - want[m]).max()
    x16 = xt.astype(np.float16)
    hip._engine.calls = 0
    a = hip.u_solve(2, 2, x16)
    hip._engine.calls = 0
    b = hip.u_solve(2, 2, x16.astype(np.float32))
    assert a.dtype == np.float32 and np.array_equal(a, b)
    clone = copy.deepcopy(hip)
    clone._engine.calls = hip._engine.calls = 0
    assert np.array_equal(clone.uz_solve(2, 2, xt), hip.uz_solve(2, 2, xt))
    assert clone.GP is not hip.GP and clone.evaluation_counter == hip.evaluation_counter


def test_reference_evaluation_counter_for_scasml():
    from scasml_gp_amd import tables
    hip, _, _ = _setup(6, 30, 10, "quad", seed=2)
    xt = _test_points(6, 8, 35)
    hip.u_solve(2, 2, xt)
    assert hip.evaluation_counter == tables.reference_evaluation_count("quad", 2, 2, True)


@pytest.mark.parametrize("d", [250, 252])
def test_large_dimension_scasml(d):
    hip, ora, _ = _setup(d, 40, 8, "quad", seed=8)
    xt = _test_points(d, 5, 36)
    got, want = hip.uz_solve(2, 2, xt), ora.uz_solve(2, 2, xt)
    assert np.all(np.abs(got - want) <= ATOL + RTOL * np.abs(want)), np.abs(got - want).max()


@pytest.mark.parametrize("B", [24, 48, 300])
def test_sample_sharded_scasml_partials(B):
    """Monte-Carlo sample sharding of the ScaSML root call: the partial sums of world = 2 add up to the
    unsharded result (un-owned rows of the point buffer are zero and their GP values unused).  B = 48: a
    workgroup of the GP evaluation spans several tree sites whose ownership alternates."""
    import torch
    hip, ora, _ = _setup(20, 60, 20, "quad", seed=6)
    xt = _test_points(20, B, 37)
    eng = hip._engine
    full, _, _ = eng.solve(3, 3, xt, stream_id=0)
    parts = [eng.solve(3, 3, xt, rank=r, world=2, stream_id=0)[0] for r in range(2)]
    for r in range(2):
        want = ora.uz_solve(3, 3, xt, rank=r, world=2, owner=eng.unit_owners(3, 3, 2)[0])
        assert np.allclose(parts[r].cpu().numpy(), want, atol=2e-4, rtol=2e-4)
    assert torch.allclose(eng.finalize_partials(parts[0] + parts[1]), full, atol=1e-4, rtol=1e-4)


def test_root_bound_is_taken_per_solve_for_converted_inputs():
    """the largest |coordinate| of the roots decides whether the fp16 planes of the evaluation can be used.  For a CPU torch
    tensor the engine works on a device COPY that is freed after the solve; the next call's copy gets the same address with version 0, and a
    bound cached under (address, size, version) was then served to roots a hundred times larger -- planes overflow, silently.  The bound is now reduced
    per solve for converted inputs (cached only for the caller's own device tensor): far-out roots handed over as a CPU tensor right after
    in-cube roots of the same shape give what the NumPy route (bound taken on the host, never cached) gives, bit for bit, and finite."""
    import torch
    hip, _, _ = _setup(20, 60, 20, "quad", seed=6)
    eng = hip._engine
    near = _test_points(20, 32, 5).astype(np.float32)
    far = near.copy()
    far[:, :-1] *= 128.0                                                   # |x| up to 64 (an exact scaling): beyond the fp16 planes' gate at d = 20 (|x| <= 49.8)
    want = eng.solve(2, 2, far, stream_id=11)[0]                           # NumPy route
    first = eng.solve(2, 2, torch.from_numpy(near), stream_id=11)[0]       # CPU tensor, in the cube: a temporary device copy
    second = eng.solve(2, 2, torch.from_numpy(far), stream_id=11)[0]       # CPU tensor, same shape, far outside
    assert bool(torch.isfinite(second).all()) and torch.equal(second, want)
    assert torch.equal(first, eng.solve(2, 2, near, stream_id=11)[0])
    dev = torch.from_numpy(far).cuda()                                     # the caller's own device tensor: cached by identity, invalidated by writes
    assert torch.equal(eng.solve(2, 2, dev, stream_id=11)[0], want) and eng._bound_cache is not None
    dev[:, :-1] /= 128.0
    assert torch.equal(eng.solve(2, 2, dev, stream_id=11)[0], first)
