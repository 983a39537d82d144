"""Oracle multilevel-Picard solvers.  TEST INFRASTRUCTURE (oracle/__init__.py).

Float64 NumPy restatement of

* ``MLP.uz_solve``               solvers/MLP.py:141-274          (variant="quad", gp=None)
* ``ScaSML.uz_solve``            solvers/ScaSML.py:149-284       (variant="quad", gp=...)
* ``MLP_full_history.uz_solve``  solvers/MLP_full_history.py:64-180   (variant="fh", gp=None)
* ``ScaSML_full_history.uz_solve`` solvers/ScaSML_full_history.py:75-199 (variant="fh", gp=...)

following SURVEY.md Appendix A.  Deliberate, documented deviations (DESIGN.md "Quirk
decisions"): random numbers are the Philox stream of oracle/philox.py keyed by the static
tree position (the reference's threefry draws and its key reuse, Appendix E-2/E-3, cannot
be reproduced); arithmetic is float64 on float32 normals (no float16 storage, E-5); the
``n == 0`` calls, whose terminal work the reference computes and discards (MLP.py:175-207,
E-8), draw nothing.  Everything deterministic is kept: tables, the stale ``delta_t`` of
MLP.py:249 (E-4), the +1e-6 guards, clipping bounds, estimator forms (E-11).

The recursion is walked path by path (sample index outermost) so every batch stays the
size of the root batch; the reference vectorises the same loops over samples
(MLP.py:215-249), which changes only the summation order.
"""
import numpy as np

from . import philox
from .tables import approx_parameters


# --------------------------------------------------------------------------- counts
def site_count(variant, n, par, tab=None):
    """Number of RNG sites (= executed path-steps) in the tree of one ``uz(n)`` call."""
    if n == 0:
        return 0
    if variant == "quad":
        Mf, Mg, Q, _, _ = tab
        s = int(Mg[par - 1, n])
        for l in range(n):
            q, mc = int(Q[par - 1, n - l - 1]), int(Mf[par - 1, n - l - 1])
            s += mc * q * (1 + site_count(variant, l, par, tab) + (site_count(variant, l - 1, par, tab) if l else 0))
        return s
    s = par ** n
    for l in range(n):
        s += par ** (n - l) * (1 + site_count(variant, l, par) + (site_count(variant, l - 1, par) if l else 0))
    return s


def reference_counts(variant, n, par, tab=None, scasml=False):
    """Work the REFERENCE performs for one root (dead n==0 terminal draws included):
    uz_solve calls, terminal jumps, Euler-Maruyama steps, f evaluations, eps_PDE
    evaluations -- the table of SURVEY.md section 3.2.  ``path_steps`` = jumps + steps."""
    if variant == "quad":
        Mf, Mg, Q, _, _ = tab
        mg = int(Mg[par - 1, n])
    else:
        mg = par ** n
    out = dict(calls=1, jumps=mg, steps=0, f=0, pde=0)
    for l in range(n):
        if variant == "quad":
            q, mc = int(Q[par - 1, n - l - 1]), int(Mf[par - 1, n - l - 1])
        else:
            q, mc = 1, par ** (n - l)
        out["steps"] += q * mc
        kids = [l] + ([l - 1] if l else [])
        for lv in kids:
            sub = reference_counts(variant, lv, par, tab, scasml)
            # the reference makes ONE vectorised call per node k (batch grown by mc)
            out["calls"] += q * sub["calls"]
            for key in ("jumps", "steps", "f", "pde"):
                out[key] += q * mc * sub[key]
            out["f"] += q * mc
        if scasml and l == 0:
            out["pde"] += q * mc
    out["path_steps"] = out["jumps"] + out["steps"]
    return out


# --------------------------------------------------------------------------- solver
class PicardOracle:
    def __init__(self, eq, variant="quad", gp=None, seed=0, stream=0, T=None, compat_crn=False, compat_f16=False, jax_stream=False):
        """compat_crn=True emulates the reference's fixed-key reuse:
        every ``uz_solve`` call draws its terminal normals from ``PRNGKey(0)`` again
        (MLP.py:167-168,178), so calls of equal shape -- the q quadrature nodes of one sample
        path -- share them; in the full-history solver the level-0 normals equal the terminal
        ones as well (MLP_full_history.py:92-93,99,138).  Used only to check the restatement
        against the relative-L2 errors logged under results*/ (tests/test_oracle_reference_band.py);
        the product path and the default oracle use independent draws."""
        self.compat_crn = bool(compat_crn)
        # jax_stream=True (quadrature variant): the normals are the REFERENCE's -- jax.random.normal(float16) under its key schedule
        # (oracle/jax_random.py; the terminal draws of every call from split(PRNGKey(0), 1)[0], the path draws from the solver's stateful
        # key, which persists across calls) -- addressed by counter, so this path-by-path walk reads the same numbers the reference's
        # batch-vectorised recursion does.  With compat_f16 the result differs from the reference's own (oracle/replay.py, which also
        # follows its float16 ARITHMETIC at the root call) by float16 roundings only: tests/test_reference_replay.py.
        self.jax_stream = bool(jax_stream)
        self.jax_splits = 0                          # path sub-keys consumed so far (the state of MLP.key, solvers/MLP.py:25, 220)
        self._jax_keys = None
        # compat_f16: the reference's solver-level float16 casts -- Equation.g / Equation.f return float16 (equations.py:261, 304),
        # ScaSML.g / ScaSML.f subtract float16 from float16 (ScaSML.py:45-47, 62), every uz_solve returns .astype(float16)
        # (MLP.py:274, ScaSML.py:284, MLP_full_history.py:180; ScaSML_full_history.py:199 does not)
        self.compat_f16 = bool(compat_f16)
        self.eq = eq
        self.variant = variant
        self.gp = gp
        self.seed, self.stream = int(seed), int(stream)
        self.d = eq.d
        self.T = eq.T if T is None else T
        self.sigma, self.mu = eq.sigma(), eq.mu()
        self.clip = eq.uncertainty if gp is not None else eq.norm_estimation
        self.sites_executed = 0

    # reference call surface -------------------------------------------------
    def uz_solve(self, n, par, x_t, root0=0, rank=0, world=1, owner=None):
        """par = rho (quad) or M (fh).  With world > 1 returns this rank's UN-CLIPPED partial
        sums: the units of the ROOT call (terminal samples, then per node (m, k) of each level's sample
        paths its "+" addend and, for l > 0, its "-" addend) are dealt to ranks by ``owner[unit]`` (round-robin, unit % world, if None;
        SURVEY.md section 8(e)); sum the ranks' results and pass them to ``finalize``."""
        x_t = np.asarray(x_t, dtype=np.float32).astype(np.float64)
        self.par = int(par)
        self.tab = approx_parameters(self.par, self.T) if self.variant == "quad" else None
        B = x_t.shape[0]
        roots = np.arange(root0, root0 + B, dtype=np.uint64)
        self._shard = (rank, world)
        self._owner = None if owner is None else np.asarray(owner)
        self.sites_executed = 0
        jx = None
        if self.jax_stream:
            jx = (self.jax_splits, np.arange(B, dtype=np.uint64))
            if self.variant == "quad":               # the full-history solvers draw everything from the one terminal key
                self.jax_splits += self._jax_splits_in_call(n)
        return self._uz(n, x_t[:, :-1].copy(), x_t[:, -1].copy(), roots, 0, top=True, cbase=0, jx=jx)

    # the reference's random stream ----------------------------------------------------------------
    def _jax_splits_in_call(self, n):
        """Sub-keys one uz_solve(n) call draws from the stateful key, its children's included (MLP.py:213-220, 231, 253)."""
        if n <= 0:
            return 0
        _, _, Q, _, _ = self.tab
        return sum(int(Q[self.par - 1, n - l - 1]) * (1 + self._jax_splits_in_call(l) + (self._jax_splits_in_call(l - 1) if l else 0))
                   for l in range(n))

    def _jax_subkey(self, i):
        from . import jax_random as jr
        if self._jax_keys is None:
            self._jax_state, self._jax_keys = jr.prng_key(0), []
            self._jax_terminal = jr.split(jr.prng_key(0), 1)[0]
        while len(self._jax_keys) <= i:
            self._jax_state, sub = jr.split(self._jax_state, 2)
            self._jax_keys.append(sub)
        return self._jax_keys[i]

    def _jax_normals(self, key, rows, width, m):
        """Sample m of a (batch, width, d) float16 draw under ``key``, for the batch rows ``rows``."""
        from . import jax_random as jr
        idx = ((rows * np.uint64(width) + np.uint64(m))[:, None] * np.uint64(self.d) + np.arange(self.d, dtype=np.uint64)[None, :])
        return jr.normal_f16_at(key, idx).astype(np.float64)

    def finalize(self, summed_partials):
        """Clip the all-reduced partial sums of a sample-sharded solve (world > 1)."""
        return np.clip(summed_partials, -self.clip, self.clip)

    def u_solve(self, n, par, x_t, **kw):
        uz = self.uz_solve(n, par, x_t, **kw)
        u = uz[:, 0:1]
        if self.gp is not None:                      # ScaSML.py:300-304
            u = u + self.gp.predict(np.asarray(x_t, dtype=np.float32).astype(np.float64))
            u = self._h(u)                           # float16 + float16 on the harness's float16 points
        return u

    # pieces -------------------------------------------------------------------
    def _g(self, X, tcol):
        P = np.concatenate([X, tcol[:, None]], axis=1)
        G = self._h(self.eq.g(P)[:, 0])
        if self.gp is not None:                      # ScaSML.py:61-63
            G = self._h(G - self.gp.predict(P)[:, 0])
        return G

    def _h(self, v):
        """.astype(float16) under compat_f16 (held in float64)."""
        if not self.compat_f16:
            return v
        with np.errstate(over="ignore"):
            return np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)

    def _f(self, X, tcol, u, z):
        P = np.concatenate([X, tcol[:, None]], axis=1)
        if self.gp is None:
            return self._h(self.eq.f(P, u[:, None], z)[:, 0])    # MLP.py:27-41
        u_hat = self.gp.predict(P)                               # ScaSML.py:43-47
        grad_x = self.gp.compute_gradient(P)[:, :-1]
        s = self.eq.sigma()
        val1 = self._h(self.eq.f(P, u[:, None] + u_hat, s * grad_x + z))
        val2 = self._h(self.eq.f(P, u_hat, s * grad_x))
        return self._h(val1 - val2)[:, 0]

    def _owned(self, top, unit):
        rank, world = self._shard
        if not top or world == 1:
            return True
        return (unit % world == rank) if self._owner is None else (int(self._owner[unit]) == rank)

    def _finish(self, u, z, top):
        out = np.concatenate([u[:, None], z], axis=1)
        if top and self._shard[1] > 1:
            return out                               # partial sums; caller reduces then clips
        c = self.clip
        # jnp.clip keeps NaN (MLP.py:274); np.clip does too
        out = np.clip(out, -c, c)
        return out if (self.variant == "fh" and self.gp is not None) else self._h(out)

    def _uz(self, n, x, t, roots, base, top=False, cbase=None, jx=None):
        """base: first RNG site of this call's subtree.  cbase: where the call's TERMINAL draws
        come from -- equal to base except under compat_crn, where it is the base the call
        would have at quadrature node k=0 of every ancestor path."""
        if n == 0:                                   # MLP.py:205-207 (dead terminal work skipped)
            return np.zeros((x.shape[0], 1 + self.d))
        if cbase is None or not self.compat_crn:
            cbase = base
        return self._uz_quad(n, x, t, roots, base, top, cbase, jx) if self.variant == "quad" \
            else self._uz_fh(n, x, t, roots, base, top, cbase, jx)

    def _terminal(self, mg, x, t, roots, base, top, eps, jx=None):
        T, d = self.T, self.d
        tau = T - t
        su = np.zeros(x.shape[0])
        sz = np.zeros((x.shape[0], d))
        for m in range(mg):                          # MLP.py:175-202
            if not self._owned(top, m):
                continue
            if jx is not None:
                self._jax_subkey(0)
                N = self._jax_normals(self._jax_terminal, jx[1], mg, m)
            else:
                N = philox.normals(self.seed, self.stream, roots, base + m, d).astype(np.float64)
            XT = x + self.mu * tau[:, None] + self.sigma * np.sqrt(tau)[:, None] * N
            G = self._g(XT, np.full_like(t, T))
            su += G
            sz += G[:, None] * N
            self.sites_executed += 1
        with np.errstate(all="ignore"):
            return su / mg, sz / (mg * (tau + eps))[:, None]

    def _uz_quad(self, n, x, t, roots, base, top, cbase, jx=None):
        Mf, Mg, Q, c, w = self.tab
        rho, T = self.par, self.T
        tau = T - t
        mg = int(Mg[rho - 1, n])
        u, z = self._terminal(mg, x, t, roots, cbase, top, 1e-6, jx)
        jsplit = jx[0] if jx is not None else 0      # sub-key index of node (l, k = 0) of this call
        o = mg
        unit = mg
        # delta_t carried across (l, k) exactly as the reference's loop nest does (MLP.py:201,249,270)
        stale = tau + 1e-6
        for l in range(n):
            q, mc = int(Q[rho - 1, n - l - 1]), int(Mf[rho - 1, n - l - 1])
            cloc = tau[:, None] * c[None, :q, q - 1] / T + t[:, None]        # MLP.py:171
            wloc = tau[:, None] * w[None, :q, q - 1] / T                     # MLP.py:172
            dts = cloc - np.concatenate([t[:, None], cloc[:, :q - 1]], axis=1)  # MLP.py:212
            s_l = site_count("quad", l, rho, self.tab)
            s_lm = site_count("quad", l - 1, rho, self.tab) if l else 0
            # per-k delta_t of the "+" term and of the "-" term
            dplus, dminus = [], []
            for k in range(q):
                own = cloc[:, k] - t + 1e-6
                if self.gp is not None:              # ScaSML.py:253 recomputes before use
                    dplus.append(own)
                else:                                # MLP.py:249 uses the stale value
                    dplus.append(stale)
                    if l:
                        stale = own                  # MLP.py:270
                dminus.append(own)
            if jx is not None and l:
                lp = l - 1                           # nodes of the previous level: q_prev x (1 + children's sub-keys)
                jsplit += int(Q[rho - 1, n - lp - 1]) * (1 + self._jax_splits_in_call(lp) + (self._jax_splits_in_call(lp - 1) if lp else 0))
            for m in range(mc):
                X = x.copy()
                W = np.zeros_like(x)
                o_k0 = o                             # offsets of this path's k=0 children (compat_crn)
                for k in range(q):
                    # sample sharding: the units are the two addends of the NODE (l, m, k): "+" (the node's term with the level-l subtree and, at
                    # l = 0, the residual term) and, for l > 0, "-" (the level-(l-1) subtree's term); the path itself (X, W) advances on every rank
                    owned = self._owned(top, unit)
                    unit += 1
                    owned_minus = False
                    if l:
                        owned_minus = self._owned(top, unit)
                        unit += 1
                    if jx is not None:
                        per_node = 1 + self._jax_splits_in_call(l) + (self._jax_splits_in_call(l - 1) if l else 0)
                        sk = jsplit + k * per_node
                        xi = self._jax_normals(self._jax_subkey(sk), jx[1], mc, m)
                        kid = (sk + 1, jx[1] * np.uint64(mc) + np.uint64(m))
                        kid2 = (sk + 1 + self._jax_splits_in_call(l), kid[1])
                    else:
                        kid = kid2 = None
                        xi = philox.normals(self.seed, self.stream, roots, base + o, self.d).astype(np.float64)
                    o += 1
                    self.sites_executed += 1
                    with np.errstate(invalid="ignore"):
                        dW = np.sqrt(dts[:, k])[:, None] * xi            # MLP.py:222
                    W = W + dW
                    X = X + self.mu * dts[:, k][:, None] + self.sigma * dW   # MLP.py:225
                    tk = cloc[:, k]
                    if not owned and not owned_minus:    # both addends of this node belong to other ranks
                        self.sites_executed -= 1
                        o += s_l + s_lm
                        continue
                    if owned:
                        sim = self._uz(l, X, tk, roots, base + o, cbase=cbase + o_k0 + 1, jx=kid)
                        y = self._f(X, tk, sim[:, 0], sim[:, 1:])
                        u = u + wloc[:, k] * y / mc                              # MLP.py:248
                        z = z + (wloc[:, k] * y)[:, None] * W / (mc * dplus[k])[:, None]   # MLP.py:249
                    o += s_l
                    if l:
                        if owned_minus:
                            sim = self._uz(l - 1, X, tk, roots, base + o, cbase=cbase + o_k0 + 1 + s_l, jx=kid2)
                            y = self._f(X, tk, sim[:, 0], sim[:, 1:])
                            u = u - wloc[:, k] * y / mc                          # MLP.py:269
                            z = z - (wloc[:, k] * y)[:, None] * W / (mc * dminus[k])[:, None]  # MLP.py:271
                        o += s_lm
                    elif self.gp is not None:                                # ScaSML.py:274-280
                        P = np.concatenate([X, tk[:, None]], axis=1)
                        eps = self.gp.compute_PDE_loss(P)[:, 0]
                        u = u + wloc[:, k] * eps / mc
                        z = z + (wloc[:, k] * eps)[:, None] * W / (mc * dminus[k])[:, None]
        return self._finish(u, z, top)

    def _uz_fh(self, n, x, t, roots, base, top, cbase, jx=None):
        M, T = self.par, self.T
        tau = T - t
        mg = M ** n
        u, z = self._terminal(mg, x, t, roots, base, top, 0.0, jx)   # MLP_full_history.py:122: no epsilon
        o = mg
        unit = mg
        for l in range(n):
            mc = M ** (n - l)                                     # MLP_full_history.py:132
            s_l = site_count("fh", l, M)
            s_lm = site_count("fh", l - 1, M) if l else 0
            for m in range(mc):
                owned = self._owned(top, unit)                     # the "+" addend of this (single-node) sample; "-" below for l > 0
                unit += 1
                owned_minus = False
                if l:
                    owned_minus = self._owned(top, unit)
                    unit += 1
                if not owned and not owned_minus:
                    o += 1 + s_l + s_lm
                    continue
                site = base + o
                o += 1
                self.sites_executed += 1
                if jx is not None:                                 # one key for the time and the normals (MLP_full_history.py:92-93, 133, 138)
                    from . import jax_random as jr
                    U = jr.uniform_f16_at(self._jax_terminal, jx[1] * np.uint64(mc) + np.uint64(m)).astype(np.float64)
                    xi = self._jax_normals(self._jax_terminal, jx[1], mc, m)
                    kid = (0, jx[1] * np.uint64(mc) + np.uint64(m))
                else:
                    kid = None
                    U = philox.uniform_tau(self.seed, self.stream, roots, site).astype(np.float64)
                    nsite = base + m if (self.compat_crn and l == 0) else site   # E-3: l=0 reuses the terminal draws
                    xi = philox.normals(self.seed, self.stream, roots, nsite, self.d).astype(np.float64)
                D = U * tau                                        # MLP_full_history.py:135
                with np.errstate(invalid="ignore"):
                    X = x + self.mu * D[:, None] + self.sigma * np.sqrt(D)[:, None] * xi   # :139-141
                    wgt = xi / np.sqrt(D + 1e-6)[:, None]          # :158-159
                tk = t + D
                if owned:
                    sim = self._uz(l, X, tk, roots, base + o, jx=kid)
                    y = self._f(X, tk, sim[:, 0], sim[:, 1:])
                    u = u + tau * y / mc                               # :157
                    z = z + (tau * y)[:, None] * wgt / mc
                o += s_l
                if l:
                    if owned_minus:
                        sim = self._uz(l - 1, X, tk, roots, base + o, jx=kid)
                        y = self._f(X, tk, sim[:, 0], sim[:, 1:])
                        u = u - tau * y / mc                           # :175
                        z = z - (tau * y)[:, None] * wgt / mc
                    o += s_lm
                elif self.gp is not None:                          # ScaSML_full_history.py:189-195
                    P = np.concatenate([X, tk[:, None]], axis=1)
                    eps = self.gp.compute_PDE_loss(P)[:, 0]
                    u = u + tau * eps / mc
                    z = z + (tau * eps)[:, None] * wgt / mc
        return self._finish(u, z, top)
