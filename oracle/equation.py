"""Oracle problem definition.  TEST INFRASTRUCTURE (oracle/__init__.py).

Restates ``Grad_Dependent_Nonlinear`` (equations/equations.py:232-417) in float64
NumPy.  Row layout is the reference's: ``x_t[:, :-1]`` spatial, ``x_t[:, -1]`` time
(solvers/MLP.py:161-162).
"""
import numpy as np


class _SemilinearBase:
    """u_t + mu sum_i d_i u + sigma^2/2 Lap u + f(u, sum_i z_i) = 0, z = sigma grad u, u(T) = g: the family the kernels cover
    (equations/equations.py:15-230 is the abstract base; f may depend on z only through sum_i z_i, mu and sigma are constants).
    Subclasses give ``f_parts(u, s) -> (f, f_u, f_s, f_uu, f_us, f_ss)``; everything else -- f, the GP's operator F and its
    derivatives -- follows."""
    eq_id = None

    def __init__(self, n_input):
        self.n_input = int(n_input)
        self.d = self.n_input - 1
        self.uncertainty = 1e-1      # equations.py:245
        self.norm_estimation = 1.0   # equations.py:246
        self.t0, self.T, self.radius = 0.0, 0.5, 0.5   # equations.py:344-357

    def sigma(self):
        return 0.25                  # equations.py:288

    def g(self, x_t):
        """terminal_constraint, equations.py:248-261 (uses the row's own time column)."""
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))))[:, None]

    def f(self, x_t, u, z):
        return self.f_parts(np.asarray(u, dtype=np.float64), np.sum(z, axis=1, keepdims=True))[0]

    def exact_solution(self, x_t):
        """equations.py:306-323."""
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))))[:, None]

    # the GP's collocation operator (models/GP.py:705-719): u_t = F(z1, z3, z5) with z1 = u, z3 = Lap u, z5 = div u
    def F_parts(self, z1, z3, z5):
        """-> F, (dF/dz1, dF/dz3, dF/dz5), (d2F/dz1^2, d2F/dz1dz5, d2F/dz5^2)."""
        s, mu = self.sigma(), self.mu()
        f, fu, fs, fuu, fus, fss = self.f_parts(z1, s * z5)
        F = -mu * z5 - (s ** 2 / 2) * z3 - f
        return F, (-fu, -(s ** 2 / 2) * np.ones_like(z1), -mu - s * fs), (-fuu, -s * fus, -s * s * fss)


class GradDependentNonlinear(_SemilinearBase):
    """``Grad_Dependent_Nonlinear`` (equations/equations.py:232-417): f = sigma u sum_i z_i, mu = -1/d - sigma^2/2."""
    eq_id = 0

    def mu(self):
        s = self.sigma()
        return -1.0 / self.d - s ** 2 / 2          # equations.py:273-276

    def f_parts(self, u, sz):
        """generator, equations.py:290-304: sigma * u * sum_i z_i."""
        s = self.sigma()
        z = np.zeros_like(u)
        return s * u * sz, s * sz, s * u, z, s + z, z


class CubicReactionDiffusion(_SemilinearBase):
    """A second member of the family, for the generic-Equation path (no reference counterpart; the reference ships one
    PDE): pure diffusion with a cubic reaction term,
        u_t + sigma^2/2 Lap u - u (1 - u) (1 + c (1 - 2u)) = 0,   c = sigma^2 d / 2,   u(T, x) = logistic(T + sum x),
    whose travelling wave u = logistic(t + sum_i x_i) is exact: with h = logistic, h' = h (1 - h), h'' = h' (1 - 2h),
    u_t + sigma^2/2 d h'' = h' (1 + c (1 - 2h)).  f does not depend on the gradient; mu = 0."""
    eq_id = 1

    def mu(self):
        return 0.0

    def f_parts(self, u, sz):
        c = self.sigma() ** 2 * self.d / 2
        w, v = u * (1 - u), 1 + c * (1 - 2 * u)
        z = np.zeros_like(u)
        return -w * v, -(1 - 2 * u) * v + 2 * c * w, z, 2 * v + 4 * c * (1 - 2 * u), z, z


class QuadraticGradientReactionDiffusion(CubicReactionDiffusion):
    """The family one step wider -- f(u, sum z, |z|^2) -- for the surrogate-free solvers (no reference counterpart):
        f = -u (1 - u) (1 + c (1 - 2u)) + (|z|^2 - sigma^2 d (u (1 - u))^2),   mu = 0,   same terminal condition.
    On the travelling wave z_i = sigma u (1 - u): the added term vanishes and logistic(t + sum x) stays exact."""
    eq_id = 2

    def f(self, x_t, u, z):
        u, z = np.asarray(u, dtype=np.float64), np.asarray(z, dtype=np.float64)
        s, d = self.sigma(), self.d
        w = u * (1 - u)
        return -w * (1 + (s * s * d / 2) * (1 - 2 * u)) + (np.sum(z * z, axis=1, keepdims=True) - s * s * d * w * w)

    def f_parts(self, u, sz):
        raise NotImplementedError("f depends on |z|^2")


def sample_points(rng, d, n_dom, n_bdy, t0=0.0, T=0.5, radius=0.5):
    """Stand-in for deepxde ``GeometryXTime.random_points / random_boundary_points``
    (equations.py:387-417): interior points uniform in the
    (d+1)-box; boundary points uniform in the cube with one random coordinate snapped to a
    face, paired with a uniform random time.  Returns float32 (n_dom, d+1), (n_bdy, d+1)."""
    dom = np.concatenate([rng.uniform(-radius, radius, (n_dom, d)), rng.uniform(t0, T, (n_dom, 1))], axis=1)
    xb = rng.uniform(-radius, radius, (n_bdy, d))
    face = rng.integers(0, d, n_bdy)
    side = rng.integers(0, 2, n_bdy) * 2 - 1
    xb[np.arange(n_bdy), face] = side * radius
    bdy = np.concatenate([xb, rng.uniform(t0, T, (n_bdy, 1))], axis=1)
    return dom.astype(np.float32), bdy.astype(np.float32)


def deepxde_points(d, n_dom, n_bdy, t0=0.0, T=0.5, radius=0.5):
    """deepxde's ``GeometryXTime.random_points`` / ``random_boundary_points`` for a Hypercube x TimeDomain geometry, call for call
    on NumPy's GLOBAL generator, in float16 (``dde.config.set_default_float("float16")``, experiment_run.py:30) -- so that
    ``np.random.seed(1234)`` (experiment_run.py:32) followed by this call is the reference's own training set, and the next call
    (or ``np.random.seed(42 + i)`` first, RepeatedExperiment.py:63-64) its test set:
      interior: the (d+1)-box in ONE draw, ``(xmax - xmin) * random((n, d+1)).astype(f16) + xmin``;
      boundary: ``x = random((n, d)).astype(f16)``; ``rand_dim = randint(d, size=n)``; that coordinate rounded to 0 or 1;
                ``(xmax - xmin) * x + xmin``; ``t = diam * random((n, 1)).astype(f16) + t0``, permuted.
    Pinned by the reference's printed "Real Solution" (tests/test_reference_logs.py).  Returns float16 arrays."""
    f16 = np.float16
    lo = np.asarray([-radius] * d + [t0], dtype=f16)
    hi = np.asarray([radius] * d + [T], dtype=f16)
    dom = (hi - lo) * np.random.random(size=(n_dom, d + 1)).astype(f16) + lo
    xb = np.random.random(size=(n_bdy, d)).astype(f16)
    pick = np.random.randint(d, size=n_bdy)
    xb[np.arange(n_bdy), pick] = np.round(xb[np.arange(n_bdy), pick])
    xb = (hi[:d] - lo[:d]) * xb + lo[:d]
    t = (f16(T - t0) * np.random.random(size=(n_bdy, 1)).astype(f16) + f16(t0)).astype(f16)
    t = np.random.permutation(t)
    return dom.astype(f16), np.hstack((xb, t)).astype(f16)


def logistic_wave_f16(x_t):
    """The reference's float16 evaluation of 1 - 1/(1 + exp(t + sum x)) (equations/equations.py:259-261, 317-323) on float16 rows:
    the sum is accumulated in float32 and rounded once (jnp.sum), every other operation is rounded to float16."""
    f16, f32 = np.float16, np.float32
    x = np.asarray(x_t, dtype=f16)
    s = x[:, :-1].astype(f32).sum(axis=1, dtype=f32).astype(f16)
    arg = (x[:, -1].astype(f32) + s.astype(f32)).astype(f16)
    with np.errstate(over="ignore"):                 # exp overflows float16 at t + sum x > 11.09 (inf, as in the reference): 1 - 1/(1 + inf) = 1
        e = np.exp(arg.astype(np.float64)).astype(f32).astype(f16)       # the correctly rounded float32 exponential, then float16: machine-independent
    q = (f32(1) / (f32(1) + e.astype(f32)).astype(f16).astype(f32)).astype(f16)
    return (f32(1) - q.astype(f32)).astype(f16)[:, None]


def rel_l2(sol, exact):
    """tests/SimpleUniform.py:110-136: NaN-masked ||sol-exact||_2 / ||exact||_2."""
    sol = np.asarray(sol, dtype=np.float64).ravel()
    exact = np.asarray(exact, dtype=np.float64).ravel()
    m = ~(np.isnan(sol) | np.isnan(exact))
    return float(np.linalg.norm(np.abs(sol[m] - exact[m])) / np.linalg.norm(exact[m]))
