"""Host-side tables and the static schedule of the Picard recursion.

The reference rebuilds these inside every recursive ``uz_solve`` call
(solvers/MLP.py:154 -> approx_parameters :111-139 -> lgwt :71-109; 76 % of its MLP wall time,
SURVEY.md section 6.4).  Here they are computed once per (rho, T) and cached, then folded
into the ``scasml_plan`` the kernels take by value.
"""
import functools

import numpy as np
from scipy.special import lambertw

from . import _lib


def _inverse_gamma(x):
    # solvers/MLP.py:57-69
    L = np.log((x + 0.036534) / np.sqrt(2 * np.pi))
    return float(np.real(L / np.real(lambertw(L / np.e)) + 0.5))


def _lgwt_reference(n_nodes, a, b, max_iter=100):
    """Nodes/weights exactly as solvers/MLP.py:71-109 produces them, including the scalar
    assignment at :99 (column 1 of the Legendre table is y[0] for every node), which makes
    the rule differ from Gauss-Legendre for n_nodes >= 2."""
    N = n_nodes - 1
    N1, N2 = N + 1, N + 2
    xu = np.linspace(-1, 1, N1)
    y = np.cos((2 * np.arange(N1) + 1) * np.pi / (2 * N + 2)) + (0.27 / N1) * np.sin(np.pi * xu * N / N2)
    y_prev = np.full(N1, 2.0)
    Lp = np.zeros(N1)
    it = 0
    with np.errstate(all="ignore"):
        while np.max(np.abs(y - y_prev)) > 2.2204e-16 and it < max_iter:
            leg = [np.ones(N1), np.full(N1, y[0])]
            for k in range(2, N1 + 1):
                leg.append(((2 * k - 1) * y * leg[k - 1] - (k - 1) * leg[k - 2]) / k)
            Lp = N2 * (leg[N1 - 1] - y * leg[N2 - 1]) / (1 - y * y)
            y_prev = y
            y = y_prev - leg[N2 - 1] / Lp
            it += 1
        x = (a * (1 - y) + b * (1 + y)) / 2
        w = (b - a) / ((1 - y * y) * (Lp * Lp)) * (N2 * N2) / (N1 * N1)
    return x, w


def inverse_gamma(gamma_input):
    """solvers/MLP.py:57-69 as the solver classes expose it: scalar or array."""
    x = np.asarray(gamma_input, dtype=np.float64)
    L = np.log((x + 0.036534) / np.sqrt(2 * np.pi))
    out = np.real(L / np.real(lambertw(L / np.e)) + 0.5)
    return float(out) if out.ndim == 0 else out


def lgwt_reference(n_nodes, a, b):
    """solvers/MLP.py:71-109 as the solver classes expose it (``lgwt(N, a, b)``)."""
    return _lgwt_reference(int(n_nodes), a, b)


@functools.lru_cache(maxsize=None)
def approx_parameters(rhomax, T=0.5):
    """(Mf, Mg, Q, c, w) of solvers/MLP.py:111-139, cached."""
    Q = np.zeros((rhomax, rhomax), dtype=np.int64)
    Mf = np.zeros((rhomax, rhomax), dtype=np.int64)
    Mg = np.zeros((rhomax, rhomax + 1), dtype=np.int64)
    for rho in range(1, rhomax + 1):
        for k in range(1, rho + 1):
            Q[rho - 1, k - 1] = int(np.round(_inverse_gamma(rho ** (k / 2))))
            Mf[rho - 1, k - 1] = int(np.round(rho ** (k / 2)))
            Mg[rho - 1, k - 1] = int(np.round(rho ** (k - 1)))
        Mg[rho - 1, rho] = rho ** rho
    qmax = int(Q.max())
    c = np.zeros((qmax, qmax))
    w = np.zeros((qmax, qmax))
    for k in range(1, qmax + 1):
        xs, ws = _lgwt_reference(k, 0.0, T)
        c[:k, k - 1] = xs[::-1]
        w[:k, k - 1] = ws[::-1]
    for arr in (Q, Mf, Mg, c, w):
        arr.setflags(write=False)
    return Mf, Mg, Q, c, w


def _levels(variant, n, par, T):
    """Per level n' <= n: terminal samples and the (q, mc, c, w) of each sub-level l."""
    out = {}
    if variant == "quad":
        Mf, Mg, Q, c, w = approx_parameters(par, T)
        for np_ in range(1, n + 1):
            terms = []
            for l in range(np_):
                q = int(Q[par - 1, np_ - l - 1])
                terms.append((q, int(Mf[par - 1, np_ - l - 1]), c[:q, q - 1] / T, w[:q, q - 1] / T))
            out[np_] = (int(Mg[par - 1, np_]), terms)
    else:
        for np_ in range(1, n + 1):
            out[np_] = (par ** np_, [(1, par ** (np_ - l), np.zeros(1), np.zeros(1)) for l in range(np_)])
    return out


def build_plan(variant, n, par, T, stale_delta_t):
    """Static schedule for uz_solve(n, par): site counts, node fractions and the delta_t
    bookkeeping of solvers/MLP.py:201,249,270 (``stale_delta_t``: MLP reuses the previous
    delta_t for the '+' term, ScaSML.py:253 recomputes it)."""
    if not 0 <= n <= _lib.MAX_LEVEL:
        raise ValueError("level n=%d outside 0..%d supported by this build" % (n, _lib.MAX_LEVEL))
    if variant == "quad" and n > par:
        raise ValueError("quadrature solver needs n <= rho (Mg table, solvers/MLP.py:175)")
    plan = _lib.Plan()
    plan.variant = 0 if variant == "quad" else 1
    plan.n = n
    levels = _levels(variant, n, par, T)
    sites = [0] * (_lib.MAX_LEVEL + 1)
    for np_ in range(1, n + 1):
        mg, terms = levels[np_]
        s = mg
        for l, (q, mc, _, _) in enumerate(terms):
            s += mc * q * (1 + sites[l] + (sites[l - 1] if l else 0))
        sites[np_] = s
    for np_ in range(1, n + 1):
        mg, terms = levels[np_]
        plan.mg[np_] = mg
        plan.sites[np_] = sites[np_]
        stale = 1.0                                     # fraction of (T - t): delta_t = T - t + 1e-6
        for l, (q, mc, cf, wf) in enumerate(terms):
            if q > _lib.MAX_Q:
                raise ValueError("q=%d quadrature nodes exceed SCASML_MAX_Q" % q)
            tm = plan.term[np_][l]
            tm.q, tm.mc = q, mc
            tm.sites_l = sites[l]
            tm.sites_lm1 = sites[l - 1] if l else 0
            for k in range(q):
                tm.cfrac[k] = cf[k]
                tm.wfrac[k] = wf[k]
                tm.dfrac[k] = cf[k] - (cf[k - 1] if k else 0.0)
                if stale_delta_t and variant == "quad":
                    tm.dplus[k] = stale
                    if l:
                        stale = cf[k]
                else:
                    tm.dplus[k] = cf[k]
    return plan


def executed_path_steps(plan):
    """Path-steps (terminal jumps + Euler-Maruyama steps) the kernels execute per root."""
    return int(plan.sites[plan.n])


def reference_path_steps(variant, n, par, T=0.5):
    """Path-steps the REFERENCE executes per root, i.e. including the terminal draws of its
    n == 0 calls whose result it discards (solvers/MLP.py:175-207)."""
    levels = _levels(variant, n, par, T) if n else {}

    def rec(np_):
        if np_ == 0:
            return 1                                    # Mg[rho-1, 0] = 1 terminal draw, discarded
        mg, terms = levels[np_]
        s = mg
        for l, (q, mc, _, _) in enumerate(terms):
            s += q * mc * (1 + rec(l) + (rec(l - 1) if l else 0))
        return s
    return rec(n)


def reference_evaluation_count(variant, n, par, scasml, T=0.5):
    """What one reference ``uz_solve`` call adds to ``evaluation_counter``
    (MLP.py:193,245,266; ScaSML.py:41,59,205,249,268; MLP_full_history.py:114,154,172;
    ScaSML_full_history.py:44,65,125,165,183 -- the last two add MC_g, as written there)."""
    levels = _levels(variant, n, par, T) if n else {}
    extra = 1 if scasml else 0                           # ScaSML.f / ScaSML.g each add 1 per call

    def rec(np_):
        mg = levels[np_][0] if np_ else 1
        c = mg + extra
        if np_ == 0:
            return c
        for l, (q, mc, _, _) in enumerate(levels[np_][1]):
            inc = mg if (variant == "fh" and scasml) else mc
            c += q * (rec(l) + inc + extra)
            if l:
                c += q * (rec(l - 1) + inc + extra)
        return c
    return rec(n)
