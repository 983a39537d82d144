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


def rel_l2(sol, exact):
    """tests/SimpleUniform.py:110-136: NaN-masked ||sol-exact||_2 / ||exact||_2."""
    sol = np.asarray(sol, dtype=np.float64).ravel()
    exact = np.asarray(exact, dtype=np.float64).ravel()
    m = ~(np.isnan(sol) | np.isnan(exact))
    return float(np.linalg.norm(np.abs(sol[m] - exact[m])) / np.linalg.norm(exact[m]))
