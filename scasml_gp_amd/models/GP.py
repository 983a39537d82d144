"""PDE-constrained Gaussian-process surrogate with the reference's call surface
(models/GP.py: ``GP`` :8-692, ``GP_Grad_Dependent_Nonlinear`` :693-769), on libscasml_hip.

What runs where:
* Gram K(phi,phi) in closed form (25 blocks, float64) ............ scasml_gp_gram      (:182-258)
* Cholesky of K + nugget*I (replaces the SVD factor, :260-267) ... scasml_cholesky
* K_p^{-1} by two blocked triangular solves, Newton step solves .. scasml_trsm_lower  (:439,533,599)
* the Newton objective as a method of its own (:430-444) ....... GP.loss_function -> scasml_gp_newton_b, scasml_trsm_lower
* the Newton iteration (:487-604): with A = K_p^{-1} explicit, gradient and Hessian of b(sol)^T A b(sol)
  are block-wise elementwise expressions (b is affine except for the product z1*z5 in F, :705-719), so
  no autodiff and no per-iteration GEMM is needed ........... scasml_gp_newton_b / _gemv / _gp_newton_system
* predict / compute_gradient / compute_PDE_loss (:653-687, 746-769) .. scasml_gp_eval, scasml_gp_gradient

Two surrogates (``compat``):
* ``None``: the operators the reference documents -- exact Laplacian features, no float16 rounding.
* ``"reference"`` (default): the surrogate the reference's code actually builds: the 5-index
  Hutchinson "Laplacian" on a cyclically shifted argument (:28-39, 87-105, 119-179) with a caller-supplied index set
  (``laplacian_idx``; the reference's comes from JAX threefry), every kernel entry rounded to float16 (:43), K_p rounded
  to float16 for the right_vector solve (:267-268, 599), z4 rounded to float16 (:719), u_hat and eps_PDE returned as float16
  values (:671, 769).  Evaluation runs on the matrix cores (csrc/gp_eval_compat_mfma.hip: the three shifted geometries are one
  x.y product against three cyclic shifts of the collocation rows) when the collocation points are exactly float16 -- the
  reference's are -- and in float64 (csrc/gp_compat.hip, ``compat_eval = "float64"``) otherwise; Gram, gradient and the CPU
  statement (oracle/gp_compat.py) are float64.  ``laplacian_idx`` may be the five indices or the name of the Threefry counter
  layout ("original" / "partitionable") with which the reference's own draw is recomputed (scasml_gp_amd/threefry.py).
* ``"reference-geometry"`` (opt-in): the SAME fit (Gram, K_p, right_vector of ``"reference"``) evaluated on the hot path without the float16
  rounding of every kernel entry, which lets the four sums factor per pair geometry (13 + 9 + 5 vector instructions per pair instead of
  27 + 13 + 13), and with the evaluation point's coordinates entering the x.y products as ONE float16 plane (half the MFMAs).  On the
  reference's own experiments (d = 20 .. 80) GP relative L2 moves by <= 1e-5 and ScaSML by <= 3e-5 against ``"reference"``
  (profiles/r04_eval_rounding_study.txt); u_hat and eps_PDE still leave as float16 values (:671, 769).
Remaining deviations in both: Newton start at 0 instead of 1e-3*N(0,1) from PRNGKey(0) (:501); Cholesky instead of
the SVD factor, so the float16 rounding of L itself (:266) has no counterpart (no measurable effect).
Which triangle: every factorisation here (scasml_cholesky, dist_gp.DistCholesky, the oracle's eigh) reads the LOWER triangle of K, i.e. it
factors tril(K) + tril(K, -1)^T.  With one rounding per entry K is symmetric and nothing is dropped.  Under ``f16_graph`` it is not (the
(dt, div) entry and its mirror are two different float16 rounding sequences, up to 1.4e-3 relative apart): the reference's SVD (:260) and
jnp.linalg.solve (:599) consume both halves, this build the lower one; kernel_phi_phi still returns the matrix as coded, asymmetric.  The GP
error on the reference's logs is unaffected to the digits they print (tests/test_gpu_f16_graph.py pins the triangle and the logged numbers).
"""
import ctypes as C
import os

import numpy as np

from .. import _lib


def _round_up(v, m):
    return (v + m - 1) // m * m


class GP(object):
    '''Gaussian Kernel Solver for high dimensional PDE'''

    def __init__(self, equation, compat="reference", laplacian_idx="partitionable", f16_graph=False):
        """compat="reference" (default): the surrogate the reference's code builds, with its own Hutchinson index draw
        (laplacian_idx: five indices, or the Threefry counter layout the draw is recomputed with -- "partitionable" reproduces the
        reference's logged errors, "original" is jax < 0.5).  compat=None: the operators the reference documents."""
        if compat == "exact":
            compat = None
        # "reference-geometry": the fit of "reference", the hot evaluation without the per-entry float16 roundings (module docstring)
        self.eval_geometry = compat == "reference-geometry"
        if self.eval_geometry:
            compat = "reference"
        if compat not in (None, "reference"):
            raise ValueError("compat must be 'reference', 'reference-geometry' or None")
        if compat == "reference" and equation.n_input - 1 < 5:
            raise ValueError("compat='reference' draws five distinct Hutchinson indices from d = %d < 5 coordinates (the reference's "
                             "random.choice(..., replace=False) fails there too); use compat=None" % (equation.n_input - 1))
        self.compat = compat
        # f16_graph (opt-in, compat="reference"): on FLOAT16 rows -- the collocation points in the fit, float16 arrays handed to predict /
        # compute_PDE_loss -- the nine Laplacian-free kernel entries follow the reference's float16 op sequence (kappa in float16 arithmetic, its
        # derivative kernels reverse-mode autodiff through it; csrc/gp_compat.hip f16_graph_blocks, oracle: OracleGPCompat(f16_graph=2)) instead of
        # one rounding per entry.  Brings the GP's relative L2 on the reference's experiments from <= 1e-4 to ~1e-5 of the logged numbers.  The
        # solvers' hot evaluation (float32 tree points) is unaffected.
        self.f16_graph = bool(f16_graph) and compat == "reference"
        self._f16_extra = 8 if os.environ.get("SCASML_GP_F16_LEVEL") == "3" else 0     # exploratory: + the Hutchinson lap blocks (round16 bit 3)
        self.laplacian_idx = None
        if compat == "reference":
            if isinstance(laplacian_idx, str):                   # the reference's own draw, models/GP.py:35
                from ..threefry import reference_laplacian_idx
                laplacian_idx = reference_laplacian_idx(equation.n_input - 1, laplacian_idx)
            idx = np.asarray(laplacian_idx if laplacian_idx is not None else [], dtype=np.int32).reshape(-1)
            if idx.size != 5 or len(set(idx.tolist())) != 5 or idx.min() < 0 or idx.max() >= equation.n_input - 1:
                raise ValueError("compat='reference' needs laplacian_idx: five distinct indices in [0, d) "
                                 "(models/GP.py:35 draws them from PRNGKey(0))")
            self.laplacian_idx = np.ascontiguousarray(idx)
        self.equation = equation
        equation.geometry()
        self.T = equation.T
        self.t0 = equation.t0
        self.n_input = equation.n_input
        self.n_output = equation.n_output
        self.d = self.n_input - 1
        self.sigma = equation.sigma() * np.sqrt(self.d)      # models/GP.py:25
        self.nugget = 1e-2                                   # :26
        self.right_vector = None
        # arithmetic of x.y in the fused evaluation: 3 = three bf16 planes on the bf16 matrix cores
        # (products exact to fp32), 22 = two fp16 planes (22-bit products, half the MFMAs), 2 = two bf16
        # planes (~2^-16 per product), 0 = fp32-input MFMA
        self.eval_split = int(os.environ.get("SCASML_GP_SPLIT", "22"))
        # compat="reference" evaluation kernel: "mfma" (matrix cores; needs float16-exact collocation points, else float64 is
        # used) or "float64" (one wavefront per point, rounding decided exactly as the NumPy statement decides it)
        self.compat_eval = os.environ.get("SCASML_GP_COMPAT_EVAL", "mfma")
        # round16 argument of the compat evaluation: bit 0 = every kernel entry rounded to float16 (:43, 55-179), bit 1 = u_hat and eps_PDE
        # leave as float16 values (:671, 769), bit 2 = one float16 plane of the evaluation point in x.y (matrix-core kernel, with bit 0 off).
        # 3 is the reference's code; 6 is compat="reference-geometry"
        self.eval_round16 = int(os.environ.get("SCASML_GP_EVAL_ROUND16", "6" if self.eval_geometry else "3"))
        self.profile = False            # bench.py: HIP-event time of every training stage into self.stage_ms
        self.stage_ms = {}

    def _stage(self, name, fn):
        """Run one training stage; with self.profile bracket it with HIP events on the launch stream."""
        if not self.profile:
            return fn()
        torch = _lib.require_gpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        e1.synchronize()
        self.stage_ms[name] = self.stage_ms.get(name, 0.0) + e0.elapsed_time(e1)
        return out

    # ------------------------------------------------------------------ device helpers
    def _points_device(self, x):
        """(n, d+1) numpy / torch -> ((n, kp) float32 CUDA rows (X, t, zero pad), was_numpy, host bound, float16 rows).  The largest
        |coordinate| of a host array is taken on the host, so that the evaluation need not read it back from the device (None for device
        tensors); ``float16 rows`` says the caller's array was float16 -- numpy or torch alike -- i.e. rows on which the reference's kernels
        are float16 arithmetic (f16_graph).  Both travel as return values: nothing about one call is parked on the instance."""
        torch = _lib.require_gpu()
        was_numpy = not isinstance(x, torch.Tensor)
        bound = None
        if was_numpy:
            f16_rows = np.asarray(x).dtype == np.float16
            arr = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
            bound = float(np.abs(arr).max()) if arr.size else 0.0
            xt = torch.from_numpy(arr).cuda()
        else:
            f16_rows = x.dtype == torch.float16
            xt = x.to(device="cuda", dtype=torch.float32)
        if xt.dim() != 2 or xt.shape[1] != self.d + 1:
            raise ValueError("points must have shape (n, %d), got %s" % (self.d + 1, tuple(xt.shape)))
        kp = int(_lib.load().scasml_point_stride(self.d))
        pts = torch.zeros((xt.shape[0], kp), dtype=torch.float32, device="cuda")
        pts[:, :self.d + 1] = xt
        return pts, was_numpy, bound, bool(f16_rows)

    def _split_for(self, x_bound):
        """eval_split, demoted from the fp16 x 2 mode to the fp32-exact bf16 x 3 mode where the coordinates may leave the fp16
        planes' range: 0.72 a |x|^2 with |x_k| <= x_bound must stay below 3e4 (include/scasml_hip.h, scasml_gp_model.x_bound)."""
        split = int(self.eval_split)
        if split == 22 and 0.7213 * (1.0 / float(self.sigma) ** 2) * x_bound * x_bound * (self.d + 1) > 3.0e4:
            return 3
        return split

    def _device_model(self, x_bound=0.0):
        if self.right_vector is None:
            raise _lib.ScasmlError("GP is not trained: call GPsolver(x_domain, x_boundary) first")
        m = _lib.GpModel()
        m.x_bound = float(x_bound)
        m.d, m.n_dom, m.n_bdy, m.n_pad = self.d, self.N_domain, self.N_boundary, self._n_pad
        m.kp = self._colloc.shape[1]
        m.split = self._split_for(x_bound) if x_bound > 0 else int(self.eval_split)
        m.a = 1.0 / float(self.sigma) ** 2
        m.sigma_eq = float(self.equation.sigma())
        m.mu_eq = float(self.equation.mu())
        m.eq_id = int(self.equation.eq_id)
        m.colloc, m.colloc_frag, m.coef = self._colloc.data_ptr(), self._frag.data_ptr(), self._coef.data_ptr()
        m.colloc_bf16 = self._bf16.data_ptr()
        m.colloc_is_f16 = int(self._colloc_is_f16)
        return m

    def _eval_device(self, pts, host_bound=None, f16_rows=False):
        """Caller-supplied points: their coordinate bound (``host_bound`` of _points_device for host arrays; one reduction + read for device
        tensors) lets far-out rows fall back to the bf16 x 3 arithmetic instead of overflowing the fp16 planes."""
        torch = _lib.require_gpu()
        out = torch.empty((pts.shape[0], 4), dtype=torch.float32, device="cuda")
        fp16_planes = (int(self.eval_split) == 22 and self.compat is None) or (self.compat == "reference" and self.compat_eval == "mfma")
        hb = host_bound
        if self.compat == "reference" and self.f16_graph and f16_rows and getattr(self, "_colloc_is_f16", False):
            # float16 rows against float16 collocation points: the float64 kernel with the reference's float16 op sequence for the Laplacian-free entries
            N = self.N_domain + self.N_boundary
            _lib.check(_lib.load().scasml_gp_eval_compat(self.d, 1.0 / float(self.sigma) ** 2, float(self.equation.sigma()), float(self.equation.mu()),
                                                         int(self.equation.eq_id), _lib.ptr(self._colloc_t), self.N_domain, self.N_boundary, N,
                                                         _lib.ptr(self._rv_dev), self.laplacian_idx.ctypes.data_as(C.c_void_p), (int(self.eval_round16) & 3) | 4 | self._f16_extra,
                                                         _lib.ptr(pts), pts.shape[0], pts.shape[1], _lib.ptr(out), None, _lib.stream_ptr()), "gp_eval_compat")
            return out
        xb = (hb if hb is not None else float(pts.abs().max())) if pts.shape[0] and fp16_planes else 0.0
        self._eval_rows(pts, pts.shape[0], 0, None, out, x_bound=max(xb, 2.0) if xb > 0 else 0.0)
        return out

    def _eval_rows(self, pts, n_rows, rows_per_site, kinds, out4, x_bound=0.0, order=None):
        """(u_hat, div u_hat, eps_PDE, dt u_hat) of the first n_rows point rows into out4: the one place the solvers and
        predict / compute_PDE_loss reach the evaluation kernels (kinds: per-site byte of scasml_plan_site_kinds or None; order: device int32
        list of the sites to evaluate, in launch order -- the as-coded matrix-core kernel then launches over those sites only)."""
        lib = _lib.load()
        if self.right_vector is None:
            raise _lib.ScasmlError("GP is not trained: call GPsolver(x_domain, x_boundary) first")
        if self.compat == "reference":
            a = 1.0 / float(self.sigma) ** 2
            xb = x_bound if x_bound > 0 else 2.0
            if self.compat_eval == "mfma" and self._compat_model is not None and 0.7213 * a * xb * xb * (self.d + 1) <= 3.0e4:
                if order is not None and kinds is not None and rows_per_site % 32 == 0 and n_rows % rows_per_site == 0:
                    _lib.check(lib.scasml_gp_eval_compat_site_list(
                        self.d, a, float(self.equation.sigma()), float(self.equation.mu()), int(self.equation.eq_id), _lib.ptr(self._compat_model),
                        self.N_domain, self.N_boundary, self.laplacian_idx.ctypes.data_as(C.c_void_p), int(self.eval_round16), float(x_bound), _lib.ptr(pts),
                        n_rows, rows_per_site, _lib.ptr(kinds), _lib.ptr(order), int(order.numel()), _lib.ptr(out4), None, _lib.stream_ptr()),
                        "gp_eval_compat_site_list")
                    return
                _lib.check(lib.scasml_gp_eval_compat_sites(
                    self.d, a, float(self.equation.sigma()), float(self.equation.mu()), int(self.equation.eq_id), _lib.ptr(self._compat_model),
                    self.N_domain, self.N_boundary, self.laplacian_idx.ctypes.data_as(C.c_void_p), int(self.eval_round16), float(x_bound), _lib.ptr(pts), n_rows,
                    rows_per_site if kinds is not None else 0, _lib.ptr(kinds) if kinds is not None else None, _lib.ptr(out4), None,
                    _lib.stream_ptr()), "gp_eval_compat_sites")
                return
            N = self.N_domain + self.N_boundary
            _lib.check(lib.scasml_gp_eval_compat(self.d, a, float(self.equation.sigma()),
                                                 float(self.equation.mu()), int(self.equation.eq_id), _lib.ptr(self._colloc_t), self.N_domain, self.N_boundary, N, _lib.ptr(self._rv_dev),
                                                 self.laplacian_idx.ctypes.data_as(C.c_void_p), int(self.eval_round16) & 3, _lib.ptr(pts), n_rows,
                                                 pts.shape[1], _lib.ptr(out4), None, _lib.stream_ptr()), "gp_eval_compat")
            return
        model = self._device_model(x_bound)
        if kinds is None:
            _lib.check(lib.scasml_gp_eval(C.byref(model), _lib.ptr(pts), n_rows, _lib.ptr(out4), None, _lib.stream_ptr()), "gp_eval")
        else:
            _lib.check(lib.scasml_gp_eval_sites(C.byref(model), _lib.ptr(pts), n_rows, rows_per_site, _lib.ptr(kinds),
                                                _lib.ptr(out4), _lib.stream_ptr()), "gp_eval")

    def _predict_device(self, x_dev):
        pts, _, hb, f16 = self._points_device(x_dev)
        return self._eval_device(pts, hb, f16)[:, 0:1]

    # ------------------------------------------------------------------ training
    def kernel_phi_phi(self, x_t_domain, x_t_boundary):
        '''K(phi,phi) + nugget*I as a CUDA float64 tensor; also factors it (models/GP.py:182-268).'''
        torch = _lib.require_gpu()
        lib = _lib.load()
        xd = torch.from_numpy(np.ascontiguousarray(np.asarray(x_t_domain), dtype=np.float32)).cuda()
        xb = torch.from_numpy(np.ascontiguousarray(np.asarray(x_t_boundary), dtype=np.float32)).cuda()
        self.N_domain, self.N_boundary = xd.shape[0], xb.shape[0]
        self.phi_dim = M = 4 * self.N_domain + self.N_boundary
        self.x_t_domain, self.x_t_boundary = np.asarray(x_t_domain), np.asarray(x_t_boundary)
        self._xd, self._xb = xd, xb
        s = _lib.stream_ptr()
        K = torch.empty((M, M), dtype=torch.float64, device="cuda")
        if self.compat == "reference":
            colloc_f16 = bool((xd.half().float() == xd).all()) and bool((xb.half().float() == xb).all())
            gram_bits = 1 | ((4 | self._f16_extra) if (self.f16_graph and colloc_f16) else 0)
            self._stage("gram", lambda: _lib.check(lib.scasml_gp_gram_compat(
                self.d, 1.0 / float(self.sigma) ** 2, _lib.ptr(xd), self.N_domain, _lib.ptr(xb), self.N_boundary,
                self.laplacian_idx.ctypes.data_as(C.c_void_p), gram_bits, _lib.ptr(K), s), "gp_gram_compat"))
        else:
            self._stage("gram", lambda: _lib.check(lib.scasml_gp_gram(
                self.d, 1.0 / float(self.sigma) ** 2, _lib.ptr(xd), self.N_domain, _lib.ptr(xb), self.N_boundary, _lib.ptr(K), s), "gp_gram"))
        Mp = _round_up(M, 32)
        L = torch.eye(Mp, dtype=torch.float64, device="cuda")
        L[:M, :M] = K
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        self._stage("cholesky", lambda: _lib.check(lib.scasml_cholesky(_lib.ptr(L), Mp, float(self.nugget), _lib.ptr(info), s), "cholesky"))
        if int(info.item()) != 0 or bool(torch.isnan(L).any()):
            raise ValueError("Cholesky decomposition resulted in NaN values.")        # models/GP.py:264-265
        self._L_pad = L
        self.cholesky_phi_phi_perturb = L[:M, :M]
        if self.compat == "reference":     # kernel_phi_phi_perturb.astype(float16) (:268): the entries are float16 already, the diagonal moves
            _lib.check(lib.scasml_round16_diag(_lib.ptr(K), M, M, float(self.nugget), s), "round16_diag")
        else:
            K.diagonal().add_(self.nugget)
        return K

    def rhs_f(self, x_t_domain):
        raise NotImplementedError

    def bdy_g(self, x_t_boundary):
        xb = np.asarray(x_t_boundary)
        if self.compat == "reference" and np.array_equal(xb.astype(np.float16).astype(xb.dtype), xb):
            xb = xb.astype(np.float16)      # the reference's boundary points are float16 arrays: g is then its float16 graph (equations.py:259-261)
        else:
            xb = xb.astype(np.float64)      # the documented operators: no float16 emulation anywhere
        return np.asarray(self.equation.g(xb), dtype=np.float64)[:, 0]   # models/GP.py:417-419

    def time_der_rep(self, sol, rhs_f):
        raise NotImplementedError

    def loss_function(self, sol, rhs_f=None, bdy_g=None, L=None):
        '''The Newton objective |L^-1 b(sol)|^2 with b = [z1, g(x_bdy), z3, F(sol) + rhs_f, z5] (models/GP.py:430-444), on the device: b by
        scasml_gp_newton_b, one blocked triangular solve against the factor of kernel_phi_phi.  ``L`` (optional): another factor of the same order.
        The reference solves ``jnp.linalg.solve(L, b)`` with its own factor, the DENSE U sqrt(S + nugget) of the SVD (:260-266, 439): a factor that
        is not lower triangular is therefore accepted too -- |L^-1 b|^2 = b^T (L L^T)^-1 b, evaluated through the Cholesky factor of L L^T (the
        triangular solve reads the lower triangle only and would silently ignore the rest).  Returns a float64 scalar (the reference casts the
        value to float16 for its log).'''
        torch = _lib.require_gpu()
        lib = _lib.load()
        if getattr(self, "N_domain", None) is None or getattr(self, "x_t_boundary", None) is None:
            raise _lib.ScasmlError("no collocation points: call kernel_phi_phi(x_domain, x_boundary) or GPsolver first (N_domain and x_t_boundary are set there)")
        if getattr(self, "_L_pad", None) is None and L is None:
            raise _lib.ScasmlError("no factor: call kernel_phi_phi(x_domain, x_boundary) or GPsolver first")
        N, Nb, M = self.N_domain, self.N_boundary, self.phi_dim
        s = _lib.stream_ptr()
        sol_d = torch.as_tensor(np.asarray(sol, dtype=np.float64).reshape(-1), device="cuda").contiguous()
        if sol_d.numel() != 3 * N:
            raise ValueError("sol has %d entries, expected 3 N_domain = %d" % (sol_d.numel(), 3 * N))
        g = self.bdy_g(self.x_t_boundary) if bdy_g is None else np.asarray(bdy_g, dtype=np.float64).reshape(-1)
        g_d = torch.as_tensor(np.asarray(g, dtype=np.float64), device="cuda").contiguous()
        if L is None:
            Lp = self._L_pad
        else:
            Lg = torch.as_tensor(np.asarray(L.detach().cpu() if isinstance(L, torch.Tensor) else L, dtype=np.float64), device="cuda")
            if Lg.shape != (M, M):
                raise ValueError("L has shape %s, expected (%d, %d)" % (tuple(Lg.shape), M, M))
            Lp = torch.eye(_round_up(M, 32), dtype=torch.float64, device="cuda")
            if bool((torch.triu(Lg, 1) != 0).any()):          # a general factor (the reference's own is U sqrt(S)): the Cholesky factor of L L^T
                Lp[:M, :M] = Lg @ Lg.T
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                _lib.check(lib.scasml_cholesky(_lib.ptr(Lp), Lp.shape[0], 0.0, _lib.ptr(info), s), "cholesky(L L^T)")
                if int(info.item()) != 0:
                    raise ValueError("L L^T is not positive definite")
            else:
                Lp[:M, :M] = Lg
        Mp = Lp.shape[0]
        b = torch.zeros((Mp, 1), dtype=torch.float64, device="cuda")
        _lib.check(lib.scasml_gp_newton_b(int(self.equation.eq_id), int(self.d), float(self.equation.sigma()), float(self.equation.mu()), _lib.ptr(sol_d),
                                          _lib.ptr(g_d), N, Nb, _lib.ptr(b), s), "gp_newton_b")
        if rhs_f is not None:
            b[2 * N + Nb:3 * N + Nb, 0] += torch.as_tensor(np.asarray(rhs_f, dtype=np.float64).reshape(-1), device="cuda")
        _lib.check(lib.scasml_trsm_lower(_lib.ptr(Lp), Mp, _lib.ptr(b), 1, 0, s), "trsm")
        return np.float64(torch.dot(b[:, 0], b[:, 0]).item())

    def _chol_solve_padded(self, Hp, rhs, n, damping):
        """(H + damping*I)^-1 rhs on an identity-padded system (in place Cholesky + two triangular solves); None if not SPD."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        npad = Hp.shape[0]
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        s = _lib.stream_ptr()
        _lib.check(lib.scasml_cholesky(_lib.ptr(Hp), npad, float(damping), _lib.ptr(info), s), "cholesky(newton)")
        if int(info.item()) != 0:
            return None
        b = torch.zeros((npad, 1), dtype=torch.float64, device="cuda")
        b[:n, 0] = rhs
        _lib.check(lib.scasml_trsm_lower(_lib.ptr(Hp), npad, _lib.ptr(b), 1, 0, s), "trsm")
        _lib.check(lib.scasml_trsm_lower(_lib.ptr(Hp), npad, _lib.ptr(b), 1, 1, s), "trsm^T")
        return b[:n, 0]

    def GPsolver(self, x_t_domain, x_t_boundary, GN_steps=20):
        '''Newton's method on b(sol)^T K_p^-1 b(sol) (models/GP.py:487-604); returns predict(x_domain).

        Host code only sequences kernels and reads two scalars per step (loss, gradient norm): Gram, Cholesky,
        K_p^-1 (two blocked triangular solves on the identity), b(sol), A b, gradient + Hessian assembly and the
        Newton solve all run in libscasml_hip (scasml_gp_gram / _cholesky / _trsm_lower / _gp_newton_b / _gemv /
        _gp_newton_system).'''
        torch = _lib.require_gpu()
        lib = _lib.load()
        if getattr(self.equation, "eq_id", None) is None or getattr(self.equation, "surrogate_free_only", False):
            raise NotImplementedError("no HIP Newton kernels for equation %s%s" % (type(self.equation).__name__, " (its f depends on |z|^2: the collocation operator of "
                                      "models/GP.py:705-719 is a function of (u, Lap u, div u) alone)" if getattr(self.equation, "eq_id", None) is not None else ""))
        Kp = self.kernel_phi_phi(x_t_domain, x_t_boundary)
        if self.compat != "reference":
            del Kp
        N, Nb, M = self.N_domain, self.N_boundary, self.phi_dim
        L = self._L_pad
        Mp = L.shape[0]
        s = _lib.stream_ptr()
        eq_id, d, sig, mu = int(self.equation.eq_id), int(self.d), float(self.equation.sigma()), float(self.equation.mu())
        A = torch.empty((Mp, Mp), dtype=torch.float64, device="cuda")   # -> K_p^-1 = L^-T L^-1
        self._stage("inverse", lambda: _lib.check(lib.scasml_cholesky_inverse(_lib.ptr(L), Mp, _lib.ptr(A), s), "cholesky_inverse"))
        bdy_g = torch.as_tensor(np.asarray(self.bdy_g(self.x_t_boundary), dtype=np.float64), device="cuda").contiguous()
        sol = torch.zeros(3 * N, dtype=torch.float64, device="cuda")
        b = torch.empty(M, dtype=torch.float64, device="cuda")
        Ab = torch.empty(M, dtype=torch.float64, device="cuda")
        grad = torch.empty(3 * N, dtype=torch.float64, device="cuda")
        npad = _round_up(3 * N, 32)
        H = torch.empty((npad, npad), dtype=torch.float64, device="cuda")
        damping = 1e-4                                                  # models/GP.py:490

        def residual(sol_):
            _lib.check(lib.scasml_gp_newton_b(eq_id, d, sig, mu, _lib.ptr(sol_), _lib.ptr(bdy_g), N, Nb, _lib.ptr(b), s), "gp_newton_b")
            _lib.check(lib.scasml_gemv(_lib.ptr(A), M, Mp, _lib.ptr(b), _lib.ptr(Ab), s), "gemv")
            return float(torch.dot(b, Ab))                              # loss = b^T A b, :430-444

        hist = [residual(sol)]
        self.grad_norms = []                                            # ||grad J|| at the start of every iteration (:518-519)
        for _ in range(GN_steps):                                       # :515-588
            _lib.check(lib.scasml_gp_newton_system(eq_id, d, sig, mu, _lib.ptr(A), Mp, N, Nb, _lib.ptr(sol), _lib.ptr(Ab),
                                                   _lib.ptr(grad), _lib.ptr(H), npad, 0, s), "gp_newton_system")
            self.grad_norms.append(float(torch.linalg.vector_norm(grad)))
            if self.grad_norms[-1] < 1e-5:                              # :521
                break
            step = self._chol_solve_padded(H, -grad, 3 * N, damping)   # :529-533 (H is overwritten by its factor)
            if step is None:
                # the full Hessian (:511) is indefinite at this iterate (large collocation sets): the reference's LU
                # solve would take the step regardless; use the Gauss-Newton part, which is positive semidefinite
                self.gauss_newton_steps = getattr(self, "gauss_newton_steps", 0) + 1
                _lib.check(lib.scasml_gp_newton_system(eq_id, d, sig, mu, _lib.ptr(A), Mp, N, Nb, _lib.ptr(sol), _lib.ptr(Ab),
                                                       _lib.ptr(grad), _lib.ptr(H), npad, 1, s), "gp_newton_system(GN)")
                step = self._chol_solve_padded(H, -grad, 3 * N, damping)
            if step is None:
                raise ValueError("Newton system is not positive definite")
            sol = sol + step                                            # alpha = 1, :541,573
            hist.append(residual(sol))
        self.loss_history = hist
        if self.compat == "reference":
            # z4 = time_der_rep(sol).astype(float16) (:719), right_vector = solve(float16(K_p), z) (:268, 599): a second
            # factorisation, of the rounded matrix (still positive definite: rounding moves only the diagonal, by < nugget)
            _lib.check(lib.scasml_round16(C.c_void_p(b.data_ptr() + 8 * (2 * N + Nb)), N, s), "round16")
            Lp = torch.eye(Mp, dtype=torch.float64, device="cuda")
            Lp[:M, :M] = Kp
            del Kp
            rv = self._chol_solve_padded(Lp, b, M, 0.0)
            del Lp
            if rv is None:
                raise ValueError("float16-rounded K_p is not positive definite")
            rv = rv.contiguous()
        else:
            rv = torch.empty(M, dtype=torch.float64, device="cuda")
            _lib.check(lib.scasml_gemv(_lib.ptr(A), M, Mp, _lib.ptr(b), _lib.ptr(rv), s), "gemv")   # right_vector = K_p^-1 z, :593-600
        self.right_vector = rv.cpu().numpy()[:, None]
        self._sol = sol
        self._pack(rv)
        return self.predict(x_t_domain)                                 # :602

    def _pack(self, rv):
        torch = _lib.require_gpu()
        lib = _lib.load()
        if self.compat == "reference":
            N = self.N_domain + self.N_boundary
            self._colloc_t = torch.empty((self.d + 1, N), dtype=torch.float64, device="cuda")
            _lib.check(lib.scasml_gp_compat_pack(self.d, _lib.ptr(self._xd), self.N_domain, _lib.ptr(self._xb), self.N_boundary,
                                                 _lib.ptr(self._colloc_t), N, _lib.stream_ptr()), "gp_compat_pack")
            self._rv_dev = rv.to(dtype=torch.float64).contiguous().clone()
            # matrix-core form of the same surrogate: needs every collocation coordinate to be exactly float16
            self._colloc_is_f16 = bool((self._xd.half().float() == self._xd).all()) and bool((self._xb.half().float() == self._xb).all())
            self._compat_model = None
            if self._colloc_is_f16:
                n_pad = _round_up(N, _lib.GP_TILE)
                self._compat_model = torch.empty((int(lib.scasml_gp_compat_model_floats(self.d, n_pad)),), dtype=torch.float32, device="cuda")
                _lib.check(lib.scasml_gp_compat_pack_mfma(self.d, 1.0 / float(self.sigma) ** 2, _lib.ptr(self._xd), self.N_domain, _lib.ptr(self._xb),
                                                          self.N_boundary, _lib.ptr(self._rv_dev), self.laplacian_idx.ctypes.data_as(C.c_void_p),
                                                          _lib.ptr(self._compat_model), _lib.stream_ptr()), "gp_compat_pack_mfma")
            torch.cuda.current_stream().synchronize()
            return
        kp = int(lib.scasml_point_stride(self.d))
        self._n_pad = _round_up(self.N_domain + self.N_boundary, _lib.GP_TILE)
        self._colloc = torch.empty((self._n_pad, kp), dtype=torch.float32, device="cuda")
        self._frag = torch.empty((self._n_pad * kp,), dtype=torch.float32, device="cuda")
        self._bf16 = torch.empty((int(lib.scasml_gp_plane_halfwords(self.d, self._n_pad)),), dtype=torch.int16, device="cuda")
        self._coef = torch.empty((int(lib.scasml_gp_coef_floats(self._n_pad)),), dtype=torch.float32, device="cuda")
        rv = rv.contiguous()
        # collocation points that are exactly fp16 (the reference's deepxde float16 arrays are) need one plane
        self._colloc_is_f16 = bool((self._xd.half().float() == self._xd).all()) and bool((self._xb.half().float() == self._xb).all())
        _lib.check(lib.scasml_gp_pack(self.d, 1.0 / float(self.sigma) ** 2, float(self.T), _lib.ptr(self._xd), self.N_domain,
                                      _lib.ptr(self._xb), self.N_boundary, _lib.ptr(rv), _lib.ptr(self._colloc),
                                      _lib.ptr(self._frag), _lib.ptr(self._bf16), _lib.ptr(self._coef), _lib.stream_ptr()), "gp_pack")
        torch.cuda.current_stream().synchronize()                  # rv may be freed by the caller

    def load_right_vector(self, x_t_domain, x_t_boundary, right_vector):
        '''Install a trained state (collocation points + right_vector) without running GPsolver.'''
        torch = _lib.require_gpu()
        self.x_t_domain, self.x_t_boundary = np.asarray(x_t_domain), np.asarray(x_t_boundary)
        self._xd = torch.from_numpy(np.ascontiguousarray(self.x_t_domain, dtype=np.float32)).cuda()
        self._xb = torch.from_numpy(np.ascontiguousarray(self.x_t_boundary, dtype=np.float32)).cuda()
        self.N_domain, self.N_boundary = self._xd.shape[0], self._xb.shape[0]
        self.phi_dim = 4 * self.N_domain + self.N_boundary
        rv = np.asarray(right_vector, dtype=np.float64).reshape(-1)
        if rv.size != self.phi_dim:
            raise ValueError("right_vector has %d entries, expected %d" % (rv.size, self.phi_dim))
        self.right_vector = rv[:, None]
        self._pack(torch.from_numpy(rv).cuda())

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        '''Everything inference needs, as NumPy arrays: collocation points, right_vector, loss history.'''
        if self.right_vector is None:
            raise _lib.ScasmlError("GP is not trained: nothing to save")
        return {"n_input": np.int64(self.n_input), "x_t_domain": np.asarray(self.x_t_domain),
                "x_t_boundary": np.asarray(self.x_t_boundary), "right_vector": np.asarray(self.right_vector),
                "loss_history": np.asarray(getattr(self, "loss_history", []), dtype=np.float64),
                "nugget": np.float64(self.nugget), "T": np.float64(self.T), "compat": np.str_(self.compat or ""), "f16_graph": np.bool_(self.f16_graph),
                # the float16 op sequence is only taken on float16 collocation points; otherwise the fit silently used one rounding per entry
                "f16_graph_effective": np.bool_(self.f16_graph and bool(getattr(self, "_colloc_is_f16", False))),
                "laplacian_idx": np.asarray(self.laplacian_idx if self.laplacian_idx is not None else [], dtype=np.int32)}

    def load_state_dict(self, state):
        if int(state["n_input"]) != self.n_input:
            raise ValueError("state is for n_input=%d, this GP has n_input=%d" % (int(state["n_input"]), self.n_input))
        if str(state.get("compat", "")) != (self.compat or "") or (
                self.compat and not np.array_equal(np.asarray(state["laplacian_idx"]), self.laplacian_idx)):
            trained = str(state.get("compat", ""))
            raise ValueError("state was trained with compat=%r, laplacian_idx=%s: construct the GP with compat=%s%s to load it" % (
                trained, state.get("laplacian_idx"), repr(trained) if trained else "None",
                (", laplacian_idx=%s" % np.asarray(state["laplacian_idx"]).tolist()) if trained else ""))
        if bool(state.get("f16_graph", False)) != bool(self.f16_graph):
            raise ValueError("state was trained with f16_graph=%s: construct the GP with the same flag" % bool(state.get("f16_graph", False)))
        self.nugget = float(state["nugget"])
        if "T" in state and float(state["T"]) != float(self.T):
            raise ValueError("state was trained with terminal time T = %g, this GP's equation has T = %g" % (float(state["T"]), float(self.T)))
        self.loss_history = list(np.asarray(state["loss_history"], dtype=np.float64))
        self.load_right_vector(state["x_t_domain"], state["x_t_boundary"], state["right_vector"])
        return self

    def save(self, path):
        np.savez_compressed(path, **self.state_dict())

    def load(self, path):
        with np.load(path) as f:
            return self.load_state_dict({k: f[k] for k in f.files})

    # ------------------------------------------------------------------ inference
    def predict(self, x_t_infer):
        '''(n, 1) posterior mean (models/GP.py:653-671).'''
        pts, was_numpy, hb, f16 = self._points_device(x_t_infer)
        out = self._eval_device(pts, hb, f16)[:, 0:1]
        return out.cpu().numpy() if was_numpy else out

    def compute_gradient(self, x_t_infer, sol_infer=None):
        '''(n, d+1) gradient of the posterior mean, time derivative last (models/GP.py:673-687).'''
        torch = _lib.require_gpu()
        lib = _lib.load()
        pts, was_numpy, _, _ = self._points_device(x_t_infer)
        grad = torch.empty((pts.shape[0], self.d + 1), dtype=torch.float32, device="cuda")
        if self.compat == "reference":      # autodiff of the as-coded u_hat (through the float16 casts), result cast to float16 (:687)
            if self.right_vector is None:
                raise _lib.ScasmlError("GP is not trained: call GPsolver(x_domain, x_boundary) first")
            N = self.N_domain + self.N_boundary
            _lib.check(lib.scasml_gp_gradient_compat(self.d, 1.0 / float(self.sigma) ** 2, _lib.ptr(self._colloc_t), self.N_domain, self.N_boundary, N,
                                                     _lib.ptr(self._rv_dev), self.laplacian_idx.ctypes.data_as(C.c_void_p), 1, _lib.ptr(pts),
                                                     pts.shape[0], pts.shape[1], _lib.ptr(grad), _lib.stream_ptr()), "gp_gradient_compat")
            return grad.cpu().numpy() if was_numpy else grad
        model = self._device_model()
        _lib.check(lib.scasml_gp_gradient(C.byref(model), _lib.ptr(pts), pts.shape[0], _lib.ptr(grad), _lib.stream_ptr()), "gp_gradient")
        return grad.cpu().numpy() if was_numpy else grad

    def compute_PDE_loss(self, x_t_infer):
        raise NotImplementedError

    # ------------------------------------------------------------------ the cross-kernel builders of the reference's class surface
    # (models/GP.py:41-179, 271-411, 630-651).  The hot path never materialises these matrices -- predict / compute_PDE_loss contract them on the
    # fly -- but callers of the reference's methods find them here, as host views over scasml_gp_cross_rows.  With compat="reference" the entries
    # are float16 VALUES (returned as float16, like the reference's .astype(jnp.float16)); with compat=None float64, not rounded.
    _OPS = {"I": 0, "lap": 1, "dt": 2, "div": 3}

    def _cross(self, op, x_t_infer, x_t_domain, x_t_boundary):
        """(N_inf, M) rows of operator `op` (or, op = 4, the (N_inf, M, d+1) gradient of the op-0 rows) against the given collocation sets."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        was_numpy = not isinstance(x_t_infer, torch.Tensor)
        f16_rows = (np.asarray(x_t_infer).dtype == np.float16) if was_numpy else x_t_infer.dtype == torch.float16

        def dev(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))
            t = t.to(device="cuda", dtype=torch.float32).contiguous().reshape(-1, self.d + 1)
            return t
        xi, xd = dev(x_t_infer), dev(x_t_domain)
        xb = dev(x_t_boundary) if x_t_boundary is not None and len(x_t_boundary) else torch.zeros((0, self.d + 1), dtype=torch.float32, device="cuda")
        nd, nb, ni = xd.shape[0], xb.shape[0], xi.shape[0]
        if nd < 1:
            raise ValueError("the cross-kernel builders need at least one domain point")
        M = 4 * nd + nb
        as_coded = self.compat == "reference"
        r16 = 1 if as_coded else 0
        if as_coded and self.f16_graph and f16_rows:
            colloc = torch.cat([xd, xb])
            if bool((colloc.half().float() == colloc).all()):          # float16 rows on float16 collocation points: the reference's float16 op sequence
                r16 |= 4 | self._f16_extra
        shape = (ni, M, self.d + 1) if op == 4 else (ni, M)
        out = torch.empty(shape, dtype=torch.float64, device="cuda")
        if ni:
            rows_per_call = 65535 * 16
            for lo in range(0, ni, rows_per_call):
                n = min(rows_per_call, ni - lo)
                _lib.check(lib.scasml_gp_cross_rows(self.d, 1.0 / float(self.sigma) ** 2, _lib.ptr(xd), nd, _lib.ptr(xb) if nb else None, nb,
                                                    self.laplacian_idx.ctypes.data_as(C.c_void_p) if as_coded else None, r16, 0 if as_coded else 1, op,
                                                    _lib.ptr(xi[lo:]), n, self.d + 1, _lib.ptr(out[lo:]), M, _lib.stream_ptr()), "gp_cross_rows")
        if as_coded:
            out = out.to(torch.float16)                               # exact: the entries are float16 values
        return out.cpu().numpy() if was_numpy else out

    def kernel_x_t_phi(self, x_t_infer, x_t_domain, x_t_boundary):
        '''K(x_t, phi): (N_infer, 4 N_domain + N_boundary), columns [kappa(dom), kappa(bdy), lap_y kappa, dt_y kappa, div_y kappa] (models/GP.py:271-294).'''
        return self._cross(0, x_t_infer, x_t_domain, x_t_boundary)

    def dx_t_kernel_x_t_phi(self, x_t_infer, x_t_domain, x_t_boundary):
        '''Gradient of K(x_t, phi) in x_t: (N_infer, 4 N_domain + N_boundary, n_input), time derivative last (models/GP.py:296-324).'''
        return self._cross(4, x_t_infer, x_t_domain, x_t_boundary)

    def laplacian_x_t_kernel_x_t_phi(self, x_t_infer, x_t_domain, x_t_boundary):
        '''The Laplacian feature rows (as coded: the 5-index Hutchinson sum on the shifted argument) (models/GP.py:326-354).'''
        return self._cross(1, x_t_infer, x_t_domain, x_t_boundary)

    def dt_x_t_kernel_x_t_phi(self, x_t_infer, x_t_domain, x_t_boundary):
        '''The time-derivative feature rows (models/GP.py:356-383).'''
        return self._cross(2, x_t_infer, x_t_domain, x_t_boundary)

    def div_x_t_kernel_x_t_phi(self, x_t_infer, x_t_domain, x_t_boundary):
        '''The divergence feature rows (models/GP.py:385-411).'''
        return self._cross(3, x_t_infer, x_t_domain, x_t_boundary)

    def kernel_x_t_phi_single(self, x_t):
        '''K(x_t, phi) for one point against the fitted collocation sets: (4 N_domain + N_boundary,) (models/GP.py:630-651).'''
        if getattr(self, "x_t_domain", None) is None:
            raise _lib.ScasmlError("no collocation points yet: call GPsolver / kernel_phi_phi / load_right_vector first")
        torch = _lib.require_gpu()
        row = x_t.reshape(1, -1) if isinstance(x_t, torch.Tensor) else np.asarray(x_t).reshape(1, -1)
        return self._cross(0, row, self.x_t_domain, self.x_t_boundary)[0]

    def _pair(self, opx, opy, x_t, y_t):
        """(L^opx_x L^opy_y kappa)(x_t, y_t) for single vectors: y_t plays a one-point domain set, whose row holds all four y-operators."""
        torch = _lib.require_gpu()
        x = x_t.reshape(1, -1) if isinstance(x_t, torch.Tensor) else np.asarray(x_t).reshape(1, -1)
        y = y_t.reshape(1, -1) if isinstance(y_t, torch.Tensor) else np.asarray(y_t).reshape(1, -1)
        if opx == "grad":
            g = self._cross(4, x, y, None)[0, 0]
            return g
        return self._cross(self._OPS[opx], x, y, None)[0, {"I": 0, "lap": 1, "dt": 2, "div": 3}[opy]]

    def kappa(self, x_t, y_t):
        '''K(x_t, y_t) for single vectors (models/GP.py:41-43).'''
        return self._pair("I", "I", x_t, y_t)

    def kappa_kernel(self, x_t, y_t):
        '''(N_x, N_y) kernel matrix (models/GP.py:45-53).'''
        return self._cross(0, x_t, y_t, None)[:, :len(y_t)]

    def dx_t_kappa(self, x_t, y_t):
        '''Gradient of kappa in x_t, (n_input,) (models/GP.py:55-57).'''
        return self._pair("grad", None, x_t, y_t)

    def dy_t_kappa(self, x_t, y_t):
        '''Gradient of kappa in y_t = -gradient in x_t (models/GP.py:65-67).'''
        return -self._pair("grad", None, x_t, y_t)


def _pair_method(name, opx, opy, cite):
    def method(self, x_t, y_t):
        return self._pair(opx, opy, x_t, y_t)
    method.__name__ = name
    method.__doc__ = "(L^%s_x L^%s_y kappa)(x_t, y_t) for single vectors (models/GP.py:%s)." % (opx, opy, cite)
    return method


# the derivative kernels of the reference's class surface, by (operator in x, operator in y): one host view each
for _name, _ox, _oy, _cite in (
        ("dt_x_t_kappa", "dt", "I", "59-63"), ("dt_y_t_kappa", "I", "dt", "69-73"), ("div_x_kappa", "div", "I", "75-79"), ("div_y_kappa", "I", "div", "81-85"),
        ("laplacian_x_t_kappa", "lap", "I", "87-95"), ("laplacian_y_t_kappa", "I", "lap", "97-105"), ("dt_x_t_dt_y_t_kappa", "dt", "dt", "107-111"),
        ("dt_x_t_div_y_kappa", "dt", "div", "113-117"), ("dt_x_t_laplacian_y_t_kappa", "dt", "lap", "119-127"), ("div_x_dt_y_t_kappa", "div", "dt", "129-133"),
        ("div_x_div_y_kappa", "div", "div", "135-139"), ("div_x_laplacian_y_t_kappa", "div", "lap", "141-149"),
        ("laplacian_x_t_dt_y_t_kappa", "lap", "dt", "151-159"), ("laplacian_x_t_div_y_kappa", "lap", "div", "161-169"),
        ("laplacian_x_t_laplacian_y_t_kappa", "lap", "lap", "171-179")):
    setattr(GP, _name, _pair_method(_name, _ox, _oy, _cite))
del _name, _ox, _oy, _cite


class GP_Semilinear(GP):
    '''The surrogate for any registered equation of the family u_t + mu div u + sigma^2/2 Lap u + f(u, sum z) = 0
    (csrc/equations.hpp): the reference has one concrete subclass per PDE (models/GP.py:693-769 for
    Grad_Dependent_Nonlinear); here the equation's eq_id selects F and f inside the kernels.'''

    def rhs_f(self, x_t):
        return np.zeros((np.asarray(x_t).shape[0],), dtype=np.float64)     # :700-702

    def time_der_rep(self, sol, rhs_f):
        '''F(z) = -mu z5 - (sigma^2/2) z3 - f(z1, sigma z5) + rhs_f; for Grad_Dependent_Nonlinear
        -sigma^2 z1 z5 + (1/d + sigma^2/2) z5 - (sigma^2/2) z3 + rhs_f  (:705-719)'''
        N = self.N_domain
        sol = np.asarray(sol, dtype=np.float64)
        return self.equation.F_parts(sol[:N], sol[N:2 * N], sol[2 * N:])[0] + rhs_f

    def DF_domain_without_time(self, sol):
        '''Jacobian of F in the unknowns (z1, z3, z5): (N, 3N) = [diag dF/dz1 | diag dF/dz3 | diag dF/dz5], float16 as the reference returns it
        (models/GP.py:722-743).  Host NumPy; the device Newton uses the same derivatives through eq_F (csrc/equations.hpp).'''
        N = self.N_domain
        sol = np.asarray(sol, dtype=np.float64).reshape(-1)
        d1, d3, d5 = self.equation.F_parts(sol[:N], sol[N:2 * N], sol[2 * N:])[1]
        return np.hstack([np.diag(np.broadcast_to(v, (N,))) for v in (d1, d3, d5)]).astype(np.float16)

    def compute_PDE_loss(self, x_t_infer):
        '''dt u + mu div u + sigma^2/2 Lap u + f(u, sigma div u); for Grad_Dependent_Nonlinear
        dt u + (sigma^2 u - 1/d - sigma^2/2) div u + sigma^2/2 Lap u  (models/GP.py:746-769)'''
        pts, was_numpy, hb, f16 = self._points_device(x_t_infer)
        out = self._eval_device(pts, hb, f16)[:, 2:3]
        return out.cpu().numpy() if was_numpy else out


class GP_Grad_Dependent_Nonlinear(GP_Semilinear):
    '''Gaussian Kernel Solver for the Grad_Dependent_Nonlinear (models/GP.py:693-769)'''


class GP_Cubic_Reaction_Diffusion(GP_Semilinear):
    '''The surrogate of equations.Cubic_Reaction_Diffusion (eq_id 1).'''
