"""GP kernels against the float64 oracle: Gram blocks, Cholesky / triangular solves, Newton
training (right_vector), fused evaluation (u_hat, div, eps_PDE, dt, Lap) and the full gradient."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(d, nd, nb, seed=0):
    from oracle.equation import GradDependentNonlinear, sample_points
    from oracle.gp import OracleGP
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    dom, bdy = sample_points(np.random.default_rng(seed), d, nd, nb)
    return GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(d + 1), compat=None), OracleGP(GradDependentNonlinear(d + 1)), dom, bdy


@pytest.mark.parametrize("d,nd,nb", [(5, 40, 9), (20, 70, 25), (100, 33, 7)])
def test_gram_matches_oracle(d, nd, nb):
    gp, ora, dom, bdy = _setup(d, nd, nb)
    K = gp.kernel_phi_phi(dom, bdy).cpu().numpy()
    want = ora.kernel_phi_phi(dom, bdy) + ora.nugget * np.eye(4 * nd + nb)
    scale = np.abs(want).max()
    assert np.abs(K - want).max() <= 1e-11 * scale
    L = gp.cholesky_phi_phi_perturb.cpu().numpy()
    assert np.allclose(np.triu(L, 1), 0) and np.abs(L @ L.T - want).max() <= 1e-10 * scale


def test_cholesky_and_trsm_against_numpy():
    import torch
    from scasml_gp_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    M = 160
    R = rng.standard_normal((M, M))
    A = R @ R.T + M * np.eye(M)
    At = torch.from_numpy(A.copy()).cuda()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.scasml_cholesky(_lib.ptr(At), M, 0.5, _lib.ptr(info), _lib.stream_ptr()), "chol")
    Lw = np.linalg.cholesky(A + 0.5 * np.eye(M))
    assert int(info.item()) == 0 and np.abs(At.cpu().numpy() - Lw).max() < 1e-10
    for nrhs in (1, 37, 160):
        Bm = rng.standard_normal((M, nrhs))
        for trans, ref in ((0, np.linalg.solve(Lw, Bm)), (1, np.linalg.solve(Lw.T, Bm))):
            Bt = torch.from_numpy(Bm.copy()).cuda()
            _lib.check(lib.scasml_trsm_lower(_lib.ptr(At), M, _lib.ptr(Bt), nrhs, trans, _lib.stream_ptr()), "trsm")
            assert np.abs(Bt.cpu().numpy() - ref).max() < 1e-9
    Ainv = torch.empty((M, M), dtype=torch.float64, device="cuda")
    _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(At), M, _lib.ptr(Ainv), _lib.stream_ptr()), "cholesky_inverse")
    want_inv = np.linalg.inv(A + 0.5 * np.eye(M))
    assert np.abs(Ainv.cpu().numpy() - want_inv).max() < 1e-11 * np.abs(want_inv).max() * M
    # not positive definite -> info reports the pivot, no exception from the C ABI
    Bad = torch.from_numpy(-np.eye(32)).cuda()
    _lib.check(lib.scasml_cholesky(_lib.ptr(Bad), 32, 0.0, _lib.ptr(info), _lib.stream_ptr()), "chol")
    assert int(info.item()) == 1
    # unsupported order is an error code + message, not a crash
    rc = lib.scasml_cholesky(_lib.ptr(At), 33, 0.0, _lib.ptr(info), _lib.stream_ptr())
    assert rc == -2 and b"multiple of 32" in lib.scasml_last_error()



def test_cholesky_lookahead_path_on_a_side_stream():
    """M >= 8192 factors with the one-panel lookahead on the call's own second stream (csrc/gp_train.hip, scasml_cholesky):
    same factor as LAPACK-on-GPU to rounding, for a size that is ragged against the 64/128/256 tilings, launched on a
    non-default stream with work queued before and after it."""
    import torch
    from scasml_gp_amd import _lib
    lib = _lib.load()
    M = 8192 + 9 * 32
    g = torch.Generator(device="cuda").manual_seed(5)
    R = torch.randn((M, 640), dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + 40.0 * torch.eye(M, dtype=torch.float64, device="cuda")
    want = torch.linalg.cholesky(A)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        L = A.clone()                                     # queued before the call on the same stream
        _lib.check(lib.scasml_cholesky(_lib.ptr(L), M, 0.0, _lib.ptr(info), side.cuda_stream), "chol")
        err = (L - want).abs().max()                      # queued after it: must see the finished factor
        Ainv = torch.empty_like(L)
        _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(L), M, _lib.ptr(Ainv), side.cuda_stream), "cholesky_inverse")
        want_inv = torch.cholesky_inverse(want)
        err_inv = (Ainv - want_inv).abs().max() / want_inv.abs().max()
    side.synchronize()
    assert int(info.item()) == 0
    assert float(err) < 1e-10 * float(want.abs().max()) * 40
    assert float(torch.triu(L, 1).abs().max()) == 0.0
    assert float(err_inv) < 1e-9          # condition number ~ 1e3 here

def test_dma_staged_update_tile_equals_the_register_staged_tile_bit_for_bit(monkeypatch):
    """The 128 x 128 FP64 update tile with LDS-DMA operand staging (csrc/f64_tile_dma.hpp) sums K in the order of the register-staged tile it
    replaces in the large trailing updates: the factor of a ragged matrix (edge tiles of 32 valid rows) and a ragged scasml_gemm_nt_sub
    (edge tiles on both sides, K not a multiple of the old 32-column chunk's pair) come out bit-identical either way, and right."""
    import torch
    from scasml_gp_amd import _lib
    lib = _lib.load()
    s = _lib.stream_ptr()
    M = 8192 + 9 * 32
    g = torch.Generator(device="cuda").manual_seed(11)
    R = torch.randn((M, 640), dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + 40.0 * torch.eye(M, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    rows, cols, K = 4096 + 96, 4096 + 224, 352
    Ap = torch.randn((rows, K + 32), dtype=torch.float64, device="cuda", generator=g)
    Bp = torch.randn((cols, K + 32), dtype=torch.float64, device="cuda", generator=g)
    C0 = torch.randn((rows, cols), dtype=torch.float64, device="cuda", generator=g)
    out = {}
    for mode in ("dma", "registers"):
        if mode == "registers":
            monkeypatch.setenv("SCASML_F64_TILE_REGISTER_STAGED", "1")
        L = A.clone()
        _lib.check(lib.scasml_cholesky(_lib.ptr(L), M, 0.0, _lib.ptr(info), s), "chol")
        Cm = C0.clone()
        _lib.check(lib.scasml_gemm_nt_sub(_lib.ptr(Cm), cols, rows, cols, _lib.ptr(Ap), K + 32, _lib.ptr(Bp), K + 32, K, 0, 0, 0, s), "gemm_nt_sub")
        Ainv = torch.empty_like(L)      # the substitution updates: k-major operands (right-hand sides; the factor transposed in the backward sweep)
        _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(L), M, _lib.ptr(Ainv), s), "cholesky_inverse")
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        out[mode] = (L, Cm, Ainv)
    monkeypatch.delenv("SCASML_F64_TILE_REGISTER_STAGED")
    assert all(torch.equal(a, b) for a, b in zip(out["dma"], out["registers"]))
    want_inv = torch.cholesky_inverse(torch.linalg.cholesky(A))
    assert float((out["dma"][2] - want_inv).abs().max()) < 1e-9 * float(want_inv.abs().max())
    want = C0 - Ap[:, :K] @ Bp[:, :K].T
    assert float((out["dma"][1] - want).abs().max()) < 1e-11 * float(want.abs().max()) * K
    assert float((out["dma"][0] - torch.linalg.cholesky(A)).abs().max()) < 1e-10 * 40 * float(A.abs().max()) ** 0.5


@pytest.mark.parametrize("d,nd,nb", [(4, 30, 10), (20, 120, 40)])
def test_training_matches_oracle(d, nd, nb):
    gp, ora, dom, bdy = _setup(d, nd, nb, seed=2)
    sol_dom = gp.GPsolver(dom, bdy, GN_steps=20)
    want_dom = ora.GPsolver(dom, bdy, GN_steps=20)
    assert len(gp.loss_history) == len(ora.loss_history)
    assert np.allclose(gp.loss_history, ora.loss_history, rtol=1e-8)
    rv, rvo = gp.right_vector, ora.right_vector
    assert rv.shape == (4 * nd + nb, 1)
    assert np.abs(rv - rvo).max() <= 1e-7 * np.abs(rvo).max()
    assert np.allclose(sol_dom, want_dom, atol=2e-5)
    # GP.loss_function (models/GP.py:430-444) as a method of its own: the Newton objective at the start, at the minimiser, with an explicit
    # boundary vector / source term / factor
    N = nd
    assert abs(gp.loss_function(np.zeros(3 * N)) - ora.loss_history[0]) <= 1e-8 * ora.loss_history[0]
    sol = gp._sol.cpu().numpy()
    assert abs(gp.loss_function(sol) - ora.loss_history[-1]) <= 1e-8 * ora.loss_history[-1]
    L = gp.cholesky_phi_phi_perturb.cpu().numpy()
    g = gp.bdy_g(bdy)
    assert abs(gp.loss_function(sol, rhs_f=np.zeros(N), bdy_g=g, L=L) - gp.loss_function(sol)) <= 1e-12 * gp.loss_function(sol)
    b = np.concatenate([sol[:N], g, sol[N:2 * N], gp.time_der_rep(sol, 0.25 * np.ones(N)), sol[2 * N:]])
    want = float(np.sum(np.linalg.solve(L, b) ** 2))
    assert abs(gp.loss_function(sol, rhs_f=0.25 * np.ones(N)) - want) <= 1e-9 * want
    with pytest.raises(ValueError):
        gp.loss_function(np.zeros(3 * N + 1))
    # the reference's own factor is the DENSE U sqrt(S + nugget) of the SVD (models/GP.py:260-266), not a triangular one: the same K_p, the same loss
    Kp = L @ L.T
    U, S, _ = np.linalg.svd(Kp)
    dense = U * np.sqrt(S)[None, :]
    assert np.abs(np.triu(dense, 1)).max() > 1e-3 and np.allclose(dense @ dense.T, Kp, rtol=1e-10, atol=1e-12)
    assert abs(gp.loss_function(sol, L=dense) - gp.loss_function(sol)) <= 1e-7 * gp.loss_function(sol)
    with pytest.raises(ValueError):
        gp.loss_function(sol, L=L[:-1, :-1])
    from scasml_gp_amd import _lib
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    with pytest.raises(_lib.ScasmlError):                     # a factor handed to an object that has no collocation points yet: a clear error, not AttributeError
        GP_Grad_Dependent_Nonlinear(gp.equation, compat=gp.compat).loss_function(sol, L=L)


@pytest.mark.parametrize("f16_colloc", [False, True])
@pytest.mark.parametrize("d,nd,nb,n_inf", [(4, 30, 10, 100), (20, 120, 40, 1000), (100, 50, 14, 77), (250, 40, 8, 40), (63, 31, 1, 65)])
def test_fused_evaluation_and_gradient_match_oracle(d, nd, nb, n_inf, f16_colloc):
    """float32 MFMA + float32 epilogue vs float64: errors scale with sum_j |kappa_j c_j|, so the
    tolerance is relative to that magnitude (1e-5) rather than to the (cancelling) result."""
    import torch
    gp, ora, dom, bdy = _setup(d, nd, nb, seed=3)
    if f16_colloc:      # float16 collocation points (the reference's deepxde arrays): the 2-MFMA fast path
        dom, bdy = dom.astype(np.float16).astype(np.float32), bdy.astype(np.float16).astype(np.float32)
    ora.GPsolver(dom, bdy, GN_steps=10)
    gp.load_right_vector(dom, bdy, ora.right_vector)
    assert gp._colloc_is_f16 == f16_colloc
    X = np.random.default_rng(4).uniform(-0.6, 0.6, (n_inf, d + 1)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1])
    mag = (np.abs(ora._features("I", X)) @ np.abs(ora.right_vector))[:, 0] + 1e-3
    u = gp.predict(X)[:, 0]
    assert np.all(np.abs(u - ora.predict(X)[:, 0]) <= 2e-5 * mag)
    eps = gp.compute_PDE_loss(X)[:, 0]
    dt, div, lap = ora.pde_parts(X)
    a = ora.a
    magp = mag * (1 + a * (1 + d))                      # derivative features carry factors of a, a*d
    assert np.all(np.abs(eps - ora.compute_PDE_loss(X)[:, 0]) <= 2e-5 * magp)
    pts = gp._points_device(X)[0]
    out4 = gp._eval_device(pts).cpu().numpy()
    assert np.all(np.abs(out4[:, 1] - div[:, 0]) <= 2e-5 * magp)
    assert np.all(np.abs(out4[:, 3] - dt[:, 0]) <= 2e-5 * magp)
    g = gp.compute_gradient(X)
    go = ora.compute_gradient(X)
    assert g.shape == (n_inf, d + 1)
    assert np.all(np.abs(g - go) <= 2e-5 * magp[:, None])
    assert np.all(np.abs(g[:, :-1].sum(1) - out4[:, 1]) <= 4e-5 * magp * np.sqrt(d))
    # torch in -> torch out, empty batch
    assert isinstance(gp.predict(torch.from_numpy(X).cuda()), torch.Tensor)
    assert gp.predict(X[:0]).shape == (0, 1)


def test_evaluation_arithmetic_modes_agree():
    """split = 3 (three bf16 planes, default) must be as exact as the fp32-input MFMA path
    (split = 0); split = 2 (two truncated planes) is the documented lower-precision fast mode."""
    gp, ora, dom, bdy = _setup(100, 150, 42, seed=6)
    ora.GPsolver(dom, bdy, GN_steps=10)
    gp.load_right_vector(dom, bdy, ora.right_vector)
    X = np.random.default_rng(7).uniform(-0.6, 0.6, (513, 101)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1])
    mag = (np.abs(ora._features("I", X)) @ np.abs(ora.right_vector))[:, 0] + 1e-3
    want = ora.predict(X)[:, 0]
    err = {}
    for split in (0, 2, 3, 22):
        gp.eval_split = split
        err[split] = np.max(np.abs(gp.predict(X)[:, 0] - want) / mag)
    assert err[3] <= 2e-6 and err[0] <= 2e-6, err
    assert err[3] <= 2 * err[0] + 2e-7, err            # fp32-exact products
    assert err[22] <= 4e-6, err                         # 22-bit products
    assert err[2] <= 1e-4, err


def test_state_dict_round_trip(tmp_path):
    gp, _, dom, bdy = _setup(12, 40, 12, seed=9)
    gp.GPsolver(dom.astype(np.float16), bdy.astype(np.float16), GN_steps=20)
    X = np.random.default_rng(1).uniform(-0.5, 0.5, (50, 13)).astype(np.float32)
    X[:, -1] = np.abs(X[:, -1])
    want = gp.predict(X)
    path = str(tmp_path / "gp.npz")
    gp.save(path)
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    fresh = GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(13), compat=None).load(path)
    assert np.array_equal(fresh.predict(X), want) and fresh._colloc_is_f16 and fresh.loss_history == gp.loss_history
    with pytest.raises(ValueError):
        GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(14), compat=None).load(path)


def test_fp16_mode_refuses_out_of_range_length_scale():
    from scasml_gp_amd import _lib
    gp, _, dom, bdy = _setup(12, 40, 12, seed=4)
    gp.GPsolver(dom.astype(np.float16), bdy.astype(np.float16), GN_steps=20)
    X = np.zeros((4, 13), dtype=np.float32)
    gp.predict(X)
    gp.sigma = 1e-3                                        # a = 1e6: k1 a^2 |x|^2 would overflow fp16
    with pytest.raises(_lib.ScasmlError):
        gp.predict(X)
    gp.eval_split = 3                                      # the bf16 x 3 mode has the fp32 exponent range (stale planes: no numbers checked)
    gp.predict(X)
