"""BASELINE.json configs[4]'s machinery past 2^31 matrix elements on ONE GPU: d = 250, 16 667 + 3 333 collocation
points -> M = 70 001 features, K(phi, phi) = 4.9e9 float64 entries (39.2 GB), 274 block rows of 256.  Everything above M = 34 999 needs
64-bit element offsets, and whole-matrix launches need more than 65 535 rows -- this file is where that is executed, not read:

* single GPU: kernel_phi_phi (Gram + two-level blocked Cholesky), both surrogates, checked through ||L L^T - K_p|| on rows sampled from the
  LAST third of the matrix (every touched element offset > 2^31) and K_p^-1 from scasml_cholesky_inverse through sampled columns of K_p K_p^-1;
* the block-row distributed path (scasml_gp_amd/dist_gp.py: Gram rows, right-looking factorisation, substitutions) at world = 1 in process
  and over 2 gloo ranks sharing the GPU, against the single-GPU Gram and factor.
* the block-row path alone at M = 140 002 (157 GB in one panel: 16 % of configs[4]'s matrix elements on one GPU), checked without a second copy.

Reference: models/GP.py:182-268 (kernel_phi_phi + factor), :593-600 (the right_vector solve)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, N_DOM, N_BDY = 250, 16667, 3333
M_XL = 4 * N_DOM + N_BDY


def _points():
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(D + 1)
    state = np.random.get_state()
    np.random.seed(1234)
    dom, bdy = eq.generate_data(N_DOM, N_BDY)
    np.random.set_state(state)
    return eq, dom, bdy


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scasml_gp_amd.dist_gp import BLK, Comm, DistCholesky
        from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
        eq, dom, bdy = _points()
        gp = GP_Grad_Dependent_Nonlinear(eq)                  # the as-coded surrogate: the estimator configs[4] runs
        ch = DistCholesky(D, 1.0 / float(gp.sigma) ** 2, dom, bdy, gp.nugget, Comm(), compat_idx=gp.laplacian_idx).build().factor()
        torch.cuda.synchronize()
        moved = ch.comm.bytes_moved
        dist.barrier()
        # the single-GPU factor, one rank at a time (each is 78 GB while it is being made)
        worst, scale = 0.0, 1.0
        for turn in range(world):
            if turn == rank:
                gp.kernel_phi_phi(dom, bdy)
                L = gp.cholesky_phi_phi_perturb
                scale = float(L.abs().max())
                for slot, i in enumerate(ch.mine):
                    r0, r1 = i * BLK, min((i + 1) * BLK, ch.M)
                    mine = torch.tril(ch.R[slot * BLK:slot * BLK + (r1 - r0), :ch.M], diagonal=r0)
                    worst = max(worst, float((mine - L[r0:r1]).abs().max()))
                del L
                gp._L_pad = gp.cholesky_phi_phi_perturb = None
                torch.cuda.empty_cache()
            dist.barrier()
        q.put((rank, worst / scale, ch.memory_bytes(), moved, len(ch.mine)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_factor_M70k_as_coded_to_the_single_gpu_factor():
    """Two gloo ranks sharing the GPU, 137 block rows each (19.7 GB per rank): the as-coded Gram rows + the right-looking factorisation with
    its broadcast / all-gather per block column give the single-GPU factor to 1e-10 on every rank's rows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1100) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, rel, mem, moved, owned in res:
        assert rel <= 1e-10, res
        assert owned == 137 and mem == 137 * 256 * 70144 * 8
        assert moved > 0.4 * 70144 * 70144 * 8 / 2          # the column panels reach every rank once


def test_block_row_factor_at_M_140k_fills_half_the_hbm():
    """The block-row path alone at d = 250, 33 334 + 6 666 collocation points: M = 140 002 features = 40 % of configs[4]'s M, 1.96e10 matrix elements
    (16 % of its 1.2e11; 157 GB of one MI355X's 288 GB in ONE panel, element offsets up to 2^34.2).  The as-coded Gram rows, the right-looking
    factorisation with look-ahead, a solve and K_p v -- checked without a second copy of the matrix: L L^T against Gram block rows saved before the
    factorisation (from the last third), and K_p (K_p^-1 b) = b."""
    import torch
    from scasml_gp_amd.dist_gp import BLK, DistCholesky
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs 200 GB of free device memory, %.0f GB are free" % (free / 1e9))
    nd, nb = 33334, 6666
    eq = Grad_Dependent_Nonlinear(D + 1)
    state = np.random.get_state()
    np.random.seed(1234)
    dom, bdy = eq.generate_data(nd, nb)
    np.random.set_state(state)
    gp = GP_Grad_Dependent_Nonlinear(eq)
    ch = DistCholesky(D, 1.0 / float(gp.sigma) ** 2, dom, bdy, gp.nugget, compat_idx=gp.laplacian_idx).build()
    M = 4 * nd + nb
    assert ch.M == M == 140002 and ch.nblk == 547 and ch.memory_bytes() == DistCholesky.budget(D, nd, nb, 1)["panel_R"] > 156e9
    picks = [546, 545, 400, 365]
    saved = {i: ch.R[i * BLK:(i + 1) * BLK].clone() for i in picks}        # K + nugget I, block rows from the last third (columns <= the block's own)
    ch.factor()
    torch.cuda.synchronize()
    scale = max(float(v.abs().max()) for v in saved.values())
    for i in picks:
        r0, r1, nc = i * BLK, min((i + 1) * BLK, M), min((i + 1) * BLK, M)
        recon = ch.R[r0:r1, :nc] @ ch.R[:nc, :nc].T                         # rows of L L^T (the panel's strict upper part is zero)
        want = saved[i][:r1 - r0, :nc]
        lower = torch.arange(nc, device="cuda")[None, :] <= torch.arange(r0, r1, device="cuda")[:, None]
        assert float(((recon - want) * lower).abs().max()) <= 1e-11 * 256 * scale, i
    del saved, recon, want
    b = torch.from_numpy(np.random.default_rng(0).standard_normal(M)).cuda()
    x = ch.solve(b)
    back = ch.matvec(x)                                                      # K_p x = L (L^T x)
    assert float((back - b).abs().max()) <= 1e-8 * float(b.abs().max()) * max(1.0, float(x.abs().max()))
    del ch
    torch.cuda.empty_cache()


@pytest.fixture(scope="module", params=["reference", None], ids=["as-coded", "documented"])
def factored(request):
    """(gp, K_p) after GP.kernel_phi_phi at M = 70 001: Gram, Cholesky factor (gp._L_pad, 70 016 x 70 016) -- 78 GB live."""
    import torch
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq, dom, bdy = _points()
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=request.param)
    Kp = gp.kernel_phi_phi(dom, bdy)
    assert gp.phi_dim == M_XL == 70001 and Kp.numel() > 2 ** 31 and gp._L_pad.shape[0] == 70016
    yield gp, Kp, dom, bdy
    del Kp
    gp._L_pad = gp.cholesky_phi_phi_perturb = None
    torch.cuda.empty_cache()


def _last_third_rows(n, seed):
    return np.sort(np.random.default_rng(seed).choice(np.arange(2 * M_XL // 3, M_XL), n, replace=False))


def test_factor_reproduces_the_gram_on_rows_beyond_2_31_elements(factored):
    import torch
    gp, Kp, _, _ = factored
    L = gp.cholesky_phi_phi_perturb
    rows_h = _last_third_rows(256, 0)
    assert int(rows_h[0]) * M_XL > 2 ** 31                  # every sampled row starts beyond the 32-bit element range
    rows = torch.from_numpy(rows_h).cuda()
    recon = L[rows] @ L.T                                   # 256 sampled rows of L L^T
    want = Kp[rows]
    scale = float(Kp.abs().max())
    cols = torch.arange(M_XL, device="cuda")[None, :]
    lower = cols < rows[:, None]                            # the factorisation reads the lower triangle
    assert float(((recon - want) * lower).abs().max()) <= 1e-11 * 128 * scale
    ar = torch.arange(256, device="cuda")
    dg = (recon[ar, rows] - want[ar, rows]).abs()
    # as coded: K_p's diagonal is float16(K_ii + nugget) (models/GP.py:268) while the Newton factor is of K + nugget I itself
    assert float(dg.max()) <= (2.0 ** -11 * 1.02 if gp.compat == "reference" else 1e-11 * 128) * scale
    assert bool(torch.isfinite(L[rows]).all()) and float(L[rows, rows].min()) > 0.0
    # nothing above the diagonal survives (zero_upper pass over 70 016 rows: grid.y strides, one workgroup row per 65 535 would not launch)
    assert float((L[rows] * (cols > rows[:, None])).abs().max()) == 0.0


def test_inverse_from_the_factor_on_columns_beyond_2_31_elements(factored):
    """scasml_cholesky_inverse at Mp = 70 016 (identity fill and mirror passes over > 65 535 rows, look-ahead chains): K_p K_p^-1 = I on sampled
    columns; K_p^-1 symmetric on sampled rows."""
    import torch
    from scasml_gp_amd import _lib
    gp, Kp, _, _ = factored
    if gp.compat == "reference":
        pytest.skip("one surrogate is enough: the inverse does not depend on how the Gram was built")
    lib = _lib.load()
    Lp = gp._L_pad
    Mp = Lp.shape[0]
    A = torch.empty((Mp, Mp), dtype=torch.float64, device="cuda")
    _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(Lp), Mp, _lib.ptr(A), _lib.stream_ptr()), "cholesky_inverse")
    rows = torch.from_numpy(_last_third_rows(128, 1)).cuda()
    early = torch.from_numpy(np.random.default_rng(2).choice(M_XL // 3, 128, replace=False)).cuda()
    for pick in (rows, early):
        prod = Kp[:, :] @ A[:M_XL, pick]                     # (M, 128): K_p is K + nugget I here (documented operators)
        eye = torch.zeros_like(prod)
        eye[pick, torch.arange(pick.numel(), device="cuda")] = 1.0
        assert float((prod - eye).abs().max()) <= 1e-6
    assert float((A[rows][:, early] - A[early][:, rows].T).abs().max()) == 0.0      # mirrored, bit for bit
    assert float((A[M_XL:, :M_XL]).abs().max()) == 0.0                              # identity padding stays decoupled
    del A
    torch.cuda.empty_cache()


def test_block_row_path_at_world_1_equals_the_single_gpu_gram_and_factor(factored):
    """DistCholesky at M = 70 001 (274 block rows, ONE 39 GB panel at world = 1): its Gram rows equal the single-GPU Gram bit for bit, its factor
    the single-GPU factor to 1e-10, its solve the single-GPU substitutions -- at element offsets up to 4.9e9."""
    import torch
    from scasml_gp_amd import _lib
    from scasml_gp_amd.dist_gp import BLK, DistCholesky
    gp, Kp, dom, bdy = factored
    L = gp.cholesky_phi_phi_perturb
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    ch = DistCholesky(D, 1.0 / float(gp.sigma) ** 2, dom, bdy, gp.nugget, compat_idx=gp.laplacian_idx).build()
    budget = DistCholesky.budget(D, N_DOM, N_BDY, 1)
    assert ch.M == M_XL and ch.nblk == 274 and ch.memory_bytes() == budget["panel_R"] == 274 * 256 * 70144 * 8 and ch.R.numel() > 2 ** 32
    for i in (273, 272, 200, 137, 1):                        # Gram rows (scasml_gp_gram_rows / _compat_rows) against the full-matrix kernels
        r0, r1 = i * BLK, min((i + 1) * BLK, M_XL)
        mine = torch.tril(ch.R[r0:r1, :M_XL], diagonal=r0 - 1)
        assert torch.equal(mine, torch.tril(Kp[r0:r1], diagonal=r0 - 1)), i
    del mine
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    ch.factor()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - before
    # the per-rank budget DESIGN.md section 6 prints for M = 350 000 is this function: what the class really held at its peak stays inside it
    assert budget["panel_R"] <= peak <= budget["total"] + (64 << 20), (peak, budget)
    scale, worst = float(L.abs().max()), 0.0
    for i in range(ch.nblk):
        r0, r1 = i * BLK, min((i + 1) * BLK, M_XL)
        worst = max(worst, float((torch.tril(ch.R[r0:r1, :M_XL], diagonal=r0) - L[r0:r1]).abs().max()))
    assert worst <= 1e-10 * scale, (worst, scale)
    lib = _lib.load()
    b = torch.from_numpy(np.random.default_rng(0).standard_normal(M_XL)).cuda()
    x = ch.solve(b)
    Mp32 = gp._L_pad.shape[0]
    ref = torch.zeros((Mp32, 1), dtype=torch.float64, device="cuda")
    ref[:M_XL, 0] = b
    _lib.check(lib.scasml_trsm_lower(_lib.ptr(gp._L_pad), Mp32, _lib.ptr(ref), 1, 0, _lib.stream_ptr()), "trsm")
    _lib.check(lib.scasml_trsm_lower(_lib.ptr(gp._L_pad), Mp32, _lib.ptr(ref), 1, 1, _lib.stream_ptr()), "trsm")
    assert float((x - ref[:M_XL, 0]).abs().max()) <= 1e-9 * float(ref.abs().max())
    got = ch.matvec(b)                                       # K_p b = L (L^T b): the preconditioner product of the Newton-CG fit
    want = L @ (L.T @ b)
    assert float((got - want).abs().max()) <= 1e-11 * float(want.abs().max())
    del ch
    torch.cuda.empty_cache()


def test_config4_solver_half_at_twenty_thousand_collocation_points():
    """BASELINE configs[4] staged, its SOLVER half: ScaSML n = rho = 3 at d = 250 on the as-coded surrogate FITTED on 16 667 + 3 333
    collocation points (M = 70 001, 19 Newton steps) -- the widest instantiation of the matrix-core evaluation kernel (16 K-steps) against 625 collocation
    tiles (a 34 MB packed model: past an XCD's L2).  Two roots against the oracle carrying the device's right_vector, every element accounted for
    (tests/_explained_parity.py); then 1024 roots through the size-independent properties.  Reference: solvers/ScaSML.py:149-284, models/GP.py:630-769."""
    import torch
    from oracle.equation import GradDependentNonlinear
    from oracle.gp_compat import OracleGPCompat
    from oracle.mlp import PicardOracle
    from _explained_parity import assert_explained
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    torch.cuda.empty_cache()
    eq, dom, bdy = _points()
    gp = GP_Grad_Dependent_Nonlinear(eq)
    gp.GPsolver(dom, bdy, GN_steps=20)
    assert gp.phi_dim == M_XL and gp._compat_model is not None and gp.loss_history[-1] < gp.loss_history[0] and gp.grad_norms[-1] < 1e-3 * gp.grad_norms[0]
    gp._L_pad = gp.cholesky_phi_phi_perturb = None           # the 39 GB factor is not needed past the fit
    torch.cuda.empty_cache()
    solver = ScaSML(eq, gp, seed=4)
    eng = solver._engine
    state = np.random.get_state()
    np.random.seed(77)
    xt = np.concatenate(eq.generate_test_data(1, 1)).astype(np.float32)
    np.random.set_state(state)
    got, _, _ = eng.solve(3, 3, xt, stream_id=3)
    got = got.cpu().numpy().astype(np.float64)
    oeq = GradDependentNonlinear(D + 1)
    ogp = OracleGPCompat(oeq, gp.laplacian_idx, round_factor=False)
    ogp.x_t_domain, ogp.x_t_boundary = np.asarray(dom, dtype=np.float64), np.asarray(bdy, dtype=np.float64)
    ogp.N_domain, ogp.N_boundary, ogp.phi_dim = N_DOM, N_BDY, M_XL
    ogp.right_vector = gp.right_vector
    want = PicardOracle(oeq, "quad", gp=ogp, seed=4, stream=3).uz_solve(3, 3, xt)
    again = assert_explained(eng, 3, 3, xt, 0, 3, want)
    assert np.array_equal(again, got) and np.abs(got[:, 0] - want[:, 0]).max() < 6e-4
    # 1024 roots (the --gp-train-xl bench leg's shape): finite, clipped at the uncertainty, deterministic, and a slice alone reproduces its rows bit for bit
    g = np.random.default_rng(1234)
    xb = torch.from_numpy(np.concatenate([g.uniform(-0.5, 0.5, (1024, D)), g.uniform(0.0, 0.5, (1024, 1))], axis=1).astype(np.float32)).cuda()
    full, uhat, _ = eng.solve(3, 3, xb, stream_id=9)
    assert full.shape == (1024, D + 1) and bool(torch.isfinite(full).all()) and bool(torch.isfinite(uhat).all())
    assert float(full.abs().max()) <= float(eq.uncertainty) * (1 + 1e-6)
    again_b, uhat_b, _ = eng.solve(3, 3, xb, stream_id=9)
    assert torch.equal(again_b, full) and torch.equal(uhat_b, uhat)
    part, uh, _ = eng.solve(3, 3, xb[500:533], root0=500, stream_id=9)
    assert torch.equal(part, full[500:533]) and torch.equal(uh, uhat[500:533])
    total = None
    for r in range(4):                                         # the north-star split over four sample ranks, walked in turn
        p, _, _ = eng.solve(3, 3, xb[:256], rank=r, world=4, stream_id=9)
        total = p.clone() if total is None else total + p
    assert torch.allclose(eng.finalize_partials(total), full[:256], atol=2e-5, rtol=1e-5)
