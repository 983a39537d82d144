"""Host driver shared by the four solver classes: builds the static schedule, owns the
device buffers and issues the HIP kernels through the C ABI.  No numerics happen here."""
import ctypes as C
import weakref

import numpy as np

from .. import _lib, tables

# cap on the point buffer of one ScaSML chunk (bytes); larger batches are walked in root chunks
POINT_BUFFER_BYTES = 24 << 30


def _as_device(x_t, torch):
    """-> (float32 contiguous CUDA tensor, was_numpy, largest |coordinate| or None).  The bound of a host array is taken on the host
    before the upload; a device tensor's is the caller's business (PicardEngine._root_bound caches it per tensor version)."""
    if isinstance(x_t, torch.Tensor):
        return x_t.to(device="cuda", dtype=torch.float32).contiguous(), False, None     # x_t itself when it already is all three
    arr = np.ascontiguousarray(np.asarray(x_t), dtype=np.float32)
    return torch.from_numpy(arr).cuda(), True, (float(np.abs(arr).max()) if arr.size else 0.0)


class PicardEngine:
    def __init__(self, equation, variant, gp=None, seed=0, compat_crn=False, compat_f16=False, compat_rng=None, reference_mode=False):
        if reference_mode:
            # ONE switch for "what the reference's solver objects compute": its random stream under its key schedule and its solver-level
            # float16 casts (the surrogate's half of it is GP(compat="reference"), the default)
            compat_rng, compat_f16 = "jax", True
        if getattr(equation, "eq_id", None) is None:
            raise NotImplementedError("no HIP kernels for equation %s (eq_id unset)" % type(equation).__name__)
        if getattr(equation, "surrogate_free_only", False) and (gp is not None or reference_mode or compat_rng == "jax"):
            raise NotImplementedError("equation %s has an f of |z|^2: its kernels are the surrogate-free Picard tree on the Philox stream (MLP, MLP_full_history); "
                                      "ScaSML would need the surrogate's full gradient at every tree site" % type(equation).__name__)
        self.equation = equation
        self.variant = variant
        self.gp = gp
        self.seed = int(seed)
        # reference key reuse (Appendix E-2/E-3) as counter keying: SCASML_RNG_COMPAT_CRN in include/scasml_hip.h
        self.compat_crn = bool(compat_crn)
        # the reference's solver-level float16 casts (g, f, every uz_solve return): SCASML_RNG_COMPAT_F16 in include/scasml_hip.h
        self.compat_f16 = bool(compat_f16)
        # compat_rng="jax": the reference's own normals -- jax.random.normal(float16) under its key schedule (SCASML_RNG_JAX_STREAM in
        # include/scasml_hip.h; unsharded solves).  The state below is the solver's ``self.key`` of solvers/MLP.py:25, 220: it
        # starts at PRNGKey(0) and every uz_solve advances it by the sub-keys that call consumes.
        if compat_rng not in (None, "jax"):
            raise ValueError("compat_rng must be None or 'jax'")
        self.compat_rng = compat_rng
        self.jax_key = (0, 0)
        self.jax_splits = 0
        self.calls = 0                 # Philox stream id: advances once per uz_solve (E-9)
        self.profile = False           # bench.py: bracket every launch with HIP events on the launch stream
        self._events = []
        self._plans = {}
        self._kinds = {}
        self._owners = {}
        self._work = {}               # point / GP-value buffers, kept across calls (4.9 GB at the headline shape)
        self._bound_cache = None      # (weakref to the caller's tensor, (address, shape, version), largest |coordinate|): _root_bound

    def __getstate__(self):               # deep-copyable (tests/ComputingBudget.py:138): drop caches and events
        st = dict(self.__dict__)
        st["_plans"], st["_events"], st["_kinds"], st["_owners"], st["_work"], st["_bound_cache"] = {}, [], {}, {}, {}, None
        return st

    def _timed(self, name, fn):
        if not self.profile:
            return fn()
        torch = _lib.require_gpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self._events.append((name, e0, e1))
        return rc

    def collect_kernel_ms(self):
        """Average HIP-event duration per kernel name since profiling was switched on."""
        torch = _lib.require_gpu()
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in self._events:
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
        self._events = []
        return {k: sum(v) / len(v) for k, v in acc.items()}

    def plan(self, n, par):
        key = (n, par)
        if key not in self._plans:
            self._plans[key] = tables.build_plan(self.variant, n, par, float(self.equation.T),
                                                 stale_delta_t=self.gp is None)
        return self._plans[key]

    def unit_owners(self, n, par, world):
        """(host uint8 owner per unit of the root call, device copy, per-rank load) from scasml_plan_deal_units: the
        Monte-Carlo units dealt by cost (longest first onto the least loaded rank)."""
        key = (n, par, world)
        if key not in self._owners:
            torch = _lib.require_gpu()
            host, load = deal_units(self.plan(n, par), world, site_cost(self.gp))
            self._owners[key] = (host, torch.from_numpy(host).cuda(), load)
        return self._owners[key]

    def site_kinds(self, n, par, rank=0, world=1):
        """Device byte per tree site: 1 where only u_hat of the surrogate is consumed, 2 where the site belongs
        to a root-call unit another rank owns (scasml_plan_site_kinds)."""
        key = (n, par, rank, world)
        if key not in self._kinds:
            torch = _lib.require_gpu()
            plan = self.plan(n, par)
            ppr = int(_lib.load().scasml_points_per_root(C.byref(plan)))
            host = np.zeros(ppr, dtype=np.uint8)
            owner = self.unit_owners(n, par, world)[0].ctypes.data_as(C.c_void_p) if world > 1 and n > 0 else None
            _lib.check(_lib.load().scasml_plan_site_kinds(C.byref(plan), rank, world, owner, host.ctypes.data_as(C.c_void_p)), "plan_site_kinds")
            self._kinds[key] = torch.from_numpy(host).cuda()
        return self._kinds[key]

    def site_order(self, n, par, rank=0, world=1):
        """Device int32 list of the sites this rank evaluates (kind != 2), by falling cost of their evaluation -- kind 0 (everything consumed),
        4 (u_hat and div), then 3 and 1 (u_hat) -- and by site index within a kind: the launch order of scasml_gp_eval_compat_site_list."""
        key = ("order", n, par, rank, world)
        if key not in self._kinds:
            torch = _lib.require_gpu()
            k = self.site_kinds(n, par, rank, world).cpu().numpy()
            rank_of = np.array([0, 2, 9, 2, 1], dtype=np.int64)          # kind -> position in the launch order
            mine = np.nonzero(k != 2)[0]
            order = mine[np.argsort(rank_of[k[mine]], kind="stable")].astype(np.int32)
            self._kinds[key] = torch.from_numpy(np.ascontiguousarray(order)).cuda()
        return self._kinds[key]

    def path_bound(self, x_max=None, plan=None):
        """Bound on |coordinate| of every tree point, the stated precondition of the fp16 evaluation planes
        (scasml_gp_model.x_bound): the largest |coordinate| of the roots actually supplied (``x_max``; the cube's half-width if not
        given) plus the largest drift plus the diffusion bound.  A tree point is reached through at most n levels of at most q_max
        Euler-Maruyama steps; every increment is sigma sqrt(dt_i) N_i with |N_i| <= 5.42 (the 24-bit inverse-CDF transform cannot
        exceed |Phi^-1(2^-25)|), so the sum is bounded by 5.42 sigma sum_i sqrt(dt_i) <= 5.42 sigma sqrt(steps * T) (Cauchy-Schwarz
        on sum dt_i <= T) -- not by a multiple of sigma sqrt(T), which bounds one normal, not a sum of truncated ones."""
        eq = self.equation
        radius = float(getattr(eq, "radius", 0.5)) if x_max is None else float(x_max)
        T = float(eq.T) - float(getattr(eq, "t0", 0.0))
        steps = 1
        if plan is not None and plan.n > 0:
            steps = plan.n * max(int(plan.term[lev][l].q) for lev in range(1, plan.n + 1) for l in range(lev))
        return max(2.0, radius + abs(float(eq.mu())) * T + 5.42 * abs(float(eq.sigma())) * (steps * T) ** 0.5)

    def problem(self):
        eq = self.equation
        p = _lib.Problem()
        p.d, p.eq_id = eq.n_input - 1, eq.eq_id
        p.T, p.mu, p.sigma = float(eq.T), float(eq.mu()), float(eq.sigma())
        p.clip = float(eq.uncertainty if self.gp is not None else eq.norm_estimation)
        return p

    def solve(self, n, par, x_t, root0=0, rank=0, world=1, stream_id=None):
        """-> (uz (B, 1+d), u_hat (B,) or None) as torch CUDA tensors, plus was_numpy."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        x, was_numpy, x_max = _as_device(x_t, torch)
        d = self.equation.n_input - 1
        if x.dim() != 2 or x.shape[1] != d + 1:
            raise ValueError("x_t must have shape (batch, %d), got %s" % (d + 1, tuple(x.shape)))
        B = x.shape[0]
        plan, prob = self.plan(n, par), self.problem()
        if self.gp is not None and float(getattr(self.gp, "T", self.equation.T)) != float(self.equation.T):
            # the GP folds its terminal time into packed row constants (site kind 3); the tree emits terminal points at the equation's T
            raise _lib.ScasmlError("the surrogate was built for terminal time T = %g, the equation now has T = %g: refit or reload the GP"
                                   % (float(self.gp.T), float(self.equation.T)))
        flags = (_lib.RNG_COMPAT_CRN if self.compat_crn else 0) | (_lib.RNG_COMPAT_F16 if self.compat_f16 else 0)
        owner = self.unit_owners(n, par, world)[1].data_ptr() if world > 1 and n > 0 else None
        jax_keys, jax_next = None, None
        if self.compat_rng == "jax" and n > 0:
            # (sample sharding: a draw is addressed by the index it has in the reference's flattened batch, which does not depend on who owns
            # the sample; every rank derives the same key words and advances its key state alike)
            if n > _lib.MAX_LEVEL:             # refuse before the key words are computed: a refused solve must not move the solver's key
                raise _lib.ScasmlError("picard_tree: level n=%d outside 1..%d" % (n, _lib.MAX_LEVEL))
            keys, jax_next = self._jax_keys(plan)
            jax_keys = keys.data_ptr()
            flags |= _lib.RNG_JAX_STREAM
        rng = _lib.Rng(self.seed, self.calls if stream_id is None else stream_id, root0, rank, world, flags, 0, owner, jax_keys)
        if stream_id is None:
            self.calls += 1
        out = torch.empty((B, d + 1), dtype=torch.float32, device="cuda")
        s = _lib.stream_ptr()
        if self.gp is None:
            _lib.check(self._timed("picard_mlp", lambda: lib.scasml_picard_tree(
                C.byref(prob), C.byref(plan), _lib.MODE_MLP, _lib.ptr(x), B, 0, rng, None, None, _lib.ptr(out), None, s)), "picard_tree")
            self._jax_commit(jax_next, stream_id)
            return out, None, was_numpy
        uhat = torch.empty((B,), dtype=torch.float32, device="cuda")
        ppr = int(lib.scasml_points_per_root(C.byref(plan)))
        kp = int(lib.scasml_point_stride(d))
        chunk = max(1, min(B, POINT_BUFFER_BYTES // (ppr * kp * 4)))
        # rows between consecutive tree sites: the chunk rounded up to the 32 rows a wavefront of the GP evaluation takes, so that
        # every such tile is ONE site (per-site choice of the cheapest epilogue, independent of how the batch is cut)
        stride = (chunk + 31) // 32 * 32
        pts, vals = self._buffers(stride * ppr, kp)   # rows of un-owned units and padding rows are never written: their (finite,
        # stale) content is evaluated or skipped by the GP kernel and never read back by ACCUMULATE
        kinds = self.site_kinds(n, par, rank, world) if n > 0 else None
        order = self.site_order(n, par, rank, world) if n > 0 else None
        # roots outside the training cube widen the bound instead of silently breaking it (host arrays: measured before the upload;
        # device tensors: one reduction per tensor version, not per solve)
        x_bound = self.path_bound(self._root_bound(x, x_max, x is x_t) if B else None, plan)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            rng_c = _lib.Rng(rng.seed, rng.stream, root0 + b0, rank, world, flags, 0, owner, jax_keys)
            xc = x[b0:b0 + nb]
            if n > 0:
                ob, ub = out[b0:b0 + nb], uhat[b0:b0 + nb]
                _lib.check(self._timed("picard_generate", lambda: lib.scasml_picard_tree(
                    C.byref(prob), C.byref(plan), _lib.MODE_GENERATE, _lib.ptr(xc), nb, stride, rng_c,
                    _lib.ptr(pts), None, None, None, s)), "picard_tree(generate)")
                self._timed("gp_eval", lambda: self.gp._eval_rows(pts, stride * ppr, stride, kinds, vals, x_bound=x_bound, order=order))
                _lib.check(self._timed("picard_accumulate", lambda: lib.scasml_picard_tree(
                    C.byref(prob), C.byref(plan), _lib.MODE_ACCUMULATE, _lib.ptr(xc), nb, stride, rng_c,
                    _lib.ptr(pts), _lib.ptr(vals), _lib.ptr(ob), _lib.ptr(ub), s)), "picard_tree(accumulate)")
            else:                          # n == 0: zeros (ScaSML.py:217-219); u_hat still needed by u_solve
                out[b0:b0 + nb].zero_()
                uhat[b0:b0 + nb] = self.gp._predict_device(xc)[:, 0]
        self._jax_commit(jax_next, stream_id)
        return out, uhat, was_numpy

    def _root_bound(self, x, x_max, own):
        """Largest |coordinate| of the roots: given for host arrays; reduced on the device otherwise.  The reduction (and the read it ends
        in) is cached ONLY for the caller's own tensor (``own``: the solve used x_t itself or a view of it), identified by a weak reference to
        the tensor that owns the storage plus (address, extent, version): while that object is alive its storage cannot be handed to
        another tensor.  A temporary made by .to() / .contiguous() (CPU, non-float32, non-contiguous input) is freed after the solve and the
        caching allocator gives the next call's temporary the same address and version 0 -- it is reduced per solve."""
        if x_max is not None:
            return x_max
        if not own:
            return float(x.abs().max())
        holder = x._base if x._base is not None else x
        key = (x.data_ptr(), tuple(x.shape), x._version)
        c = self._bound_cache
        if c is None or c[0]() is not holder or c[1] != key:
            self._bound_cache = c = (weakref.ref(holder), key, float(x.abs().max()))
        return c[2]

    def _jax_keys(self, plan):
        """Device words [terminal key | the sub-keys this solve draws from the solver's stateful key] and the key state AFTER the solve,
        which the caller commits (_jax_commit) once every launch of the solve has been accepted."""
        from .. import threefry
        torch = _lib.require_gpu()
        q = [[int(plan.term[level][l].q) for l in range(level)] for level in range(plan.n + 1)]
        words, key_after = threefry.solver_key_words(q, plan.n, self.jax_key, quadrature=self.variant == "quad")
        self._work["jax_keys"] = torch.from_numpy(words.view(np.int32).reshape(-1).copy()).cuda()
        return self._work["jax_keys"], (key_after, len(words) - 1)

    def _jax_commit(self, jax_next, stream_id):
        """Advance the solver's key (``self.key`` of solvers/MLP.py:220) by what the finished solve drew.  A solve that raised has not moved
        it; a replay of an earlier call (explicit ``stream_id``) reads the current key and does not move it either -- as it leaves the Philox
        call counter alone."""
        if jax_next is not None and stream_id is None:
            self.jax_key, self.jax_splits = jax_next[0], self.jax_splits + jax_next[1]

    def _buffers(self, rows, kp):
        """The site-major point buffer and its GP values: owned by the engine and reused by later calls of the same
        shape, so a step allocates nothing (the caching allocator hid this after the first call; a first call did not)."""
        torch = _lib.require_gpu()
        have = self._work.get("pts")
        if have is None or have.shape[0] < rows or have.shape[1] != kp:
            self._work.pop("pts", None)
            self._work.pop("vals", None)
            self._work["pts"] = torch.zeros((rows, kp), dtype=torch.float32, device="cuda")    # zeroed once: rows nobody writes stay finite
            self._work["vals"] = torch.zeros((rows, 4), dtype=torch.float32, device="cuda")
        return self._work["pts"][:rows], self._work["vals"][:rows]

    def finalize_partials(self, summed):
        """Clip all-reduced partial sums of a sample-sharded solve and, under compat_f16, apply the root call's ``.astype(float16)``
        (MLP.py:272-274, ScaSML.py:282-284, MLP_full_history.py:178-180; ScaSML_full_history.py:196-199 does not cast): the kernel leaves
        both to the reduction's end when the root call is sharded."""
        lib = _lib.load()
        round16 = 1 if (self.compat_f16 and not (self.variant == "fh" and self.gp is not None)) else 0
        _lib.check(lib.scasml_clip_round16(_lib.ptr(summed), summed.numel(), self.problem().clip, round16, _lib.stream_ptr()), "clip")
        return summed

    def evaluation_increment(self, n, par):
        return tables.reference_evaluation_count(self.variant, n, par, self.gp is not None, float(self.equation.T))


# What a tree site costs a rank, relative to an Euler-Maruyama site of a level-0 term (all of the surrogate's outputs consumed): {that site,
# a level l > 0 site (u_hat and div u_hat only), a terminal site (u_hat only), one replayed path step} -- scasml_plan_deal_units' site_cost_h.
# Measured at the headline shape on one MI355X (profiles/r06_sample_sharding_rank_times.txt): the evaluation kernel alone with every site declared of
# one kind, plus 3.4 us of GENERATE + ACCUMULATE per site of 16384 roots.  No surrogate: every site is one path step.
SITE_COST = {"reference": (1.0, 0.62, 0.50, 0.04),            # gp_eval_compat_mfma (as coded): 37.1 / 21.8 / 17.0 us per site
             "reference-geometry": (1.0, 0.62, 0.50, 0.06),   # the same geometries per kind, cheaper epilogues
             "documented": (1.0, 0.75, 0.50, 0.10),           # gp_eval_bf16: one geometry, the epilogue shrinks with what is consumed
             None: (1.0, 1.0, 1.0, 0.5)}


def site_cost(gp):
    if gp is None:
        return SITE_COST[None]
    if gp.compat == "reference":
        return SITE_COST["reference-geometry" if getattr(gp, "eval_geometry", False) else "reference"]
    return SITE_COST["documented"]


def deal_units(plan, world, cost=None):
    """Host side of scasml_plan_deal_units: (owner uint8 per unit, load per rank); cost: the four relative site costs (site_cost) or None for
    the ABI's defaults."""
    lib = _lib.load()
    n = plan.n
    units = int(plan.mg[n]) + sum(int(plan.term[n][l].mc) * int(plan.term[n][l].q) * (2 if l else 1) for l in range(n))   # terminal samples + the addends of the nodes
    owner = np.zeros(max(units, 1), dtype=np.uint8)
    load = np.zeros(world, dtype=np.float64)
    w = np.ascontiguousarray(cost, dtype=np.float64) if cost is not None else None
    if w is not None and w.shape != (4,):
        raise ValueError("site costs: four numbers (level-0 site, level l > 0 site, terminal site, replayed step)")
    got = lib.scasml_plan_deal_units(C.byref(plan), world, w.ctypes.data_as(C.c_void_p) if w is not None else None,
                                     owner.ctypes.data_as(C.c_void_p), units, load.ctypes.data_as(C.c_void_p))
    if got != units:
        raise _lib.ScasmlError("plan_deal_units failed (%d): %s" % (got, lib.scasml_last_error().decode()))
    return owner[:units].copy(), load


def deliver(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t
