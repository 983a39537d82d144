"""Replay of the reference's own runs of ``MLP.u_solve``, digit for digit.  TEST INFRASTRUCTURE (oracle/__init__.py).

oracle/mlp.py restates the multilevel-Picard estimator in float64 on Philox normals; that is what the HIP path is checked against.  THIS
module answers the other question -- is that restatement the reference's algorithm? -- by walking the reference's code path with the
reference's own random stream (oracle/jax_random.py) and the dtype every one of its operations has under JAX's promotion rules with x64 on
(experiment_run.py:46), and comparing with what its runs printed.  What had to be got right (solvers/MLP.py:141-274):

* keys: the terminal draws of EVERY call come from ``split(PRNGKey(0), 1)[0]`` (:167-168, 178: the key is rebuilt in every call), the path
  draws from the solver's stateful ``self.key`` (:220), which children advance before their parent's next node;
* the call for the whole batch is ONE array draw: a child call sees the batch flattened (root, sample) row-major (:231, 253);
* ``n == 0`` calls return zeros after their (discarded) terminal work (:205-207); f(x, 0, 0) = 0 for this equation, so at n = rho = 2 only
  the level-1 nodes contribute, each through a terminal-only child estimate;
* dtypes at the ROOT call, whose x_t is float16 (experiment_run.py:30): ``T - t``, its square root, ``dW``, the drift and the terminal
  points are float16 operations, rounded one by one (:179-181); ``jnp.full(.., T)`` is weakly typed, so the terminal inputs stay float16
  and ``terminal_constraint`` runs its float16 graph (:185; equations.py:259-261); ``jnp.mean`` of float16 accumulates in float32 and rounds
  once; ``cloc`` / ``wloc`` are float64 products of the float16 ``T - t`` (:171-172); ``X`` becomes float64 at its first update (:225);
* dtypes in child calls (float64 x_t): float64 arithmetic on float16 normals; g, f, and every return are ``.astype(float16)`` (:274;
  equations.py:261, 304); ``equation.f`` on the float16 child estimate is float16 arithmetic (:241; equations.py:303);
* the stale ``delta_t`` of the z estimator (:201, 249, 270) -- which does not reach u.

tests/test_reference_replay.py: ``MLP rel L2`` and the L1 statistics of results/Grad_Dependent_Nonlinear/{20,40,60,80}d/SimpleUniform/
SimpleUniform.log, all printed digits.
"""
import numpy as np

from . import jax_random as jr
from .equation import logistic_wave_f16
from .tables import approx_parameters

F16, F32, F64 = np.float16, np.float32, np.float64


def _mean16(a, axis):
    """``jnp.mean`` / of a float16 array: float32 accumulation and division, one rounding."""
    return (np.sum(a.astype(F32), axis=axis, dtype=F32) / F32(a.shape[axis])).astype(F16)


def _sum16(a, axis):
    return np.sum(a.astype(F32), axis=axis, keepdims=True, dtype=F32).astype(F16)


class ReplayMLP:
    """``solvers.MLP.MLP`` on ``Grad_Dependent_Nonlinear``: the object's key state persists across calls, as the harness's solver2 does."""

    def __init__(self, eq):
        self.eq, self.d, self.T = eq, eq.d, eq.T
        self.sigma, self.mu = eq.sigma(), eq.mu()
        self.key = jr.prng_key(0)                                  # MLP.py:25
        self.terminal_key = jr.split(jr.prng_key(0), 1)[0]         # MLP.py:167-168
        self.splits = 0

    def _next_subkey(self):                                        # MLP.py:220
        self.key, sub = jr.split(self.key, 2)
        self.splits += 1
        return sub

    def _g(self, rows):
        """Equation.g on float64 rows (float64 graph, one cast) or float16 rows (float16 graph)."""
        if rows.dtype == F16:
            return logistic_wave_f16(rows)
        with np.errstate(over="ignore"):
            return (1 - 1 / (1 + np.exp(rows[:, -1] + np.sum(rows[:, :-1], axis=1)))).astype(F16)[:, None]

    def _f(self, child):
        """Equation.f(x, u, z) = sigma u sum z on a child's (u, z) columns; x does not enter."""
        u, z = child[:, 0:1], child[:, 1:]
        if child.dtype == F16:
            return ((F16(self.sigma) * u).astype(F16) * _sum16(z, 1)).astype(F16)
        return (self.sigma * u * np.sum(z, axis=1, keepdims=True)).astype(F16)

    def uz_solve(self, n, rho, x_t):
        Mf, Mg, Q, c, w = approx_parameters(rho, self.T)
        T, d, B = self.T, self.d, x_t.shape[0]
        half = x_t.dtype == F16
        x, t = x_t[:, :-1], x_t[:, -1]
        tau = (F16(T) - t).astype(F16) if half else T - t
        tau64, t64 = tau.astype(F64), t.astype(F64)
        cloc = tau64[:, None, None] * c[None] / T + t64[:, None, None]              # :171
        wloc = tau64[:, None, None] * w[None] / T                                    # :172
        mg = int(Mg[rho - 1, n])
        N = jr.normal_f16(self.terminal_key, (B, mg, d))                              # :178
        if half:
            dW = (np.sqrt(tau).astype(F16)[:, None, None] * N).astype(F16)           # :179
            moved = (x[:, None, :] + (F16(self.mu) * tau).astype(F16)[:, None, None]).astype(F16)
            XT = (moved + (F16(self.sigma) * dW).astype(F16)).astype(F16)             # :181
            rows = np.concatenate([XT, np.full((B, mg, 1), T, dtype=F16)], axis=2)    # :185 (weakly typed fill)
            eps = F16(1e-6)
        else:
            XT = x[:, None, :] + (self.mu * tau)[:, None, None] + self.sigma * (np.sqrt(tau)[:, None, None] * N.astype(F64))
            rows = np.concatenate([XT, np.full((B, mg, 1), T)], axis=2)
            eps = 1e-6
        G = self._g(rows.reshape(-1, d + 1)).reshape(B, mg, 1)                        # :191
        u = _mean16(G, 1)                                                             # :198
        delta = (tau + eps).astype(tau.dtype)[:, None]                                # :200
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            z = _mean16((G * N).astype(F16), 1) / delta                               # :201 (float16 / float16, or float16 / float64)
        if n == 0:                                                                    # :205-207
            return np.zeros((B, 1 + d), dtype=F16 if half else F64)
        u, z = u.astype(F64), z.astype(F64)          # both are float64 after their first update below; the values carry over exactly
        first = True
        for l in range(n):
            q, mc = int(Q[rho - 1, n - l - 1]), int(Mf[rho - 1, n - l - 1])
            steps = cloc[:, :q, q - 1] - np.concatenate([t64[:, None], cloc[:, :q - 1, q - 1]], axis=1)      # :212
            X = np.repeat(x.astype(F64)[:, None, :], mc, axis=1)
            W = np.zeros((B, mc, d))
            for k in range(q):
                xi = jr.normal_f16(self._next_subkey(), (B, mc, d)).astype(F64)       # :220-221
                with np.errstate(invalid="ignore"):
                    dW = np.sqrt(steps[:, k])[:, None, None] * xi                     # :222
                W = W + dW
                X = X + (self.mu * steps[:, k][:, None, None] + self.sigma * dW)      # :225
                node = np.concatenate([X, np.repeat(cloc[:, k, q - 1][:, None, None], mc, axis=1)], axis=2).reshape(-1, d + 1)
                y = self._f(self.uz_solve(l, rho, node)).reshape(B, mc, 1)           # :234-243
                wk = wloc[:, k, q - 1][:, None]
                scale = (F16(mc) * delta).astype(F16).astype(F64) if (half and first) else mc * delta.astype(F64)
                with np.errstate(divide="ignore", invalid="ignore"):
                    u = u + wk * _mean16(y, 1).astype(F64)                            # :248
                    z = z + wk * np.sum(y.astype(F64) * W, axis=1) / scale            # :249 (delta_t as last assigned)
                if l:
                    y = self._f(self.uz_solve(l - 1, rho, node)).reshape(B, mc, 1)   # :256-266
                    delta = (cloc[:, k, q - 1] - t64 + 1e-6)[:, None]                 # :270
                    first = False
                    with np.errstate(divide="ignore", invalid="ignore"):
                        u = u - wk * _mean16(y, 1).astype(F64)                        # :269
                        z = z - wk * np.sum(y.astype(F64) * W, axis=1) / (mc * delta)  # :271
        bound = self.eq.norm_estimation
        return np.clip(np.concatenate([u, z], axis=1), -bound, bound).astype(F16)     # :272-274

    def u_solve(self, n, rho, x_t):                                                   # :276-288
        return self.uz_solve(n, rho, np.asarray(x_t))[:, 0:1]


class ReplayMLPFullHistory:
    """``solvers.MLP_full_history.MLP_full_history`` (solvers/MLP_full_history.py:64-196).  Every draw of every call -- the terminal normals,
    the uniform times and the path normals of every level -- comes from the ONE key ``split(PRNGKey(0), 1)[0]`` (:92-93, 99, 133, 138): the time
    ``tau`` and the normals of a sample are functions of overlapping threefry outputs, and calls of equal shape repeat each other's numbers.
    With a float16 root batch nothing ever promotes (there are no float64 tables here): the whole recursion is float16 arithmetic."""

    def __init__(self, eq):
        self.eq, self.d, self.T = eq, eq.d, eq.T
        self.sigma, self.mu = F16(eq.sigma()), F16(eq.mu())
        self.key = jr.split(jr.prng_key(0), 1)[0]

    @staticmethod
    def _f(sigma, child):
        return ((sigma * child[:, 0:1]).astype(F16) * _sum16(child[:, 1:], 1)).astype(F16)

    def uz_solve(self, n, M, x_t):
        assert x_t.dtype == F16
        T, d, B = F16(self.T), self.d, x_t.shape[0]
        x, t = x_t[:, :-1], x_t[:, -1]
        tau = (T - t).astype(F16)
        mg = M ** n
        N = jr.normal_f16(self.key, (B, mg, d))                                                        # :99
        dW = (np.sqrt(tau).astype(F16)[:, None, None] * N).astype(F16)
        XT = ((x[:, None, :] + (self.mu * tau).astype(F16)[:, None, None]).astype(F16) + (self.sigma * dW).astype(F16)).astype(F16)   # :103
        G = logistic_wave_f16(np.concatenate([XT, np.full((B, mg, 1), T, dtype=F16)], axis=2).reshape(-1, d + 1)).reshape(B, mg, 1)
        u = _mean16(G, 1)                                                                              # :120
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            z = (_mean16((G * N).astype(F16), 1) / tau[:, None]).astype(F16)                           # :122-123
        if n == 0:
            return np.zeros((B, 1 + d), dtype=F16)                                                     # :126-128
        for l in range(n):
            mc = M ** (n - l)
            step = (jr.uniform_f16(self.key, (B, mc)) * tau[:, None]).astype(F16)[:, :, None]          # :133-135
            xi = jr.normal_f16(self.key, (B, mc, d))                                                   # :138
            dW = (np.sqrt(step).astype(F16) * xi).astype(F16)                                          # :139
            X = (x[:, None, :] + ((self.mu * step).astype(F16) + (self.sigma * dW).astype(F16)).astype(F16)).astype(F16)   # :141
            node = np.concatenate([X, (t[:, None, None] + step).astype(F16)], axis=2).reshape(-1, d + 1)   # :145
            root = np.sqrt((step + F16(1e-6)).astype(F16)).astype(F16)                                 # :158
            for sign, level in ((1, l), (-1, l - 1)):
                if level < 0 or (sign < 0 and not l):
                    continue
                y = self._f(self.sigma, self.uz_solve(level, M, node)).reshape(B, mc, 1)               # :147-155 / :166-173
                with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
                    du = (tau[:, None] * _mean16(y, 1)).astype(F16)                                    # :157 / :175
                    dz = (tau[:, None] * _mean16(((y * xi).astype(F16) / root).astype(F16), 1)).astype(F16)   # :159 / :177
                    u = (u + du).astype(F16) if sign > 0 else (u - du).astype(F16)
                    z = (z + dz).astype(F16) if sign > 0 else (z - dz).astype(F16)
        bound = F16(self.eq.norm_estimation)
        return np.clip(np.concatenate([u, z], axis=1), -bound, bound).astype(F16)                      # :178-180

    def u_solve(self, n, M, x_t):
        return self.uz_solve(n, M, np.asarray(x_t))[:, 0:1]
