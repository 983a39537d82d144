"""Oracle problem definition.  TEST INFRASTRUCTURE (oracle/__init__.py).

Restates ``Grad_Dependent_Nonlinear`` (equations/equations.py:232-417) in float64
NumPy.  Row layout is the reference's: ``x_t[:, :-1]`` spatial, ``x_t[:, -1]`` time
(solvers/MLP.py:161-162).
"""
import numpy as np


class GradDependentNonlinear:
    def __init__(self, n_input):
        self.n_input = int(n_input)
        self.d = self.n_input - 1
        self.uncertainty = 1e-1      # equations.py:245
        self.norm_estimation = 1.0   # equations.py:246
        self.t0, self.T, self.radius = 0.0, 0.5, 0.5   # equations.py:344-357

    def sigma(self):
        return 0.25                  # equations.py:288

    def mu(self):
        s = self.sigma()
        return -1.0 / self.d - s ** 2 / 2          # equations.py:273-276

    def g(self, x_t):
        """terminal_constraint, equations.py:248-261 (uses the row's own time column)."""
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))))[:, None]

    def f(self, x_t, u, z):
        """generator, equations.py:290-304: sigma * u * sum_i z_i."""
        return self.sigma() * u * np.sum(z, axis=1, keepdims=True)

    def exact_solution(self, x_t):
        """equations.py:306-323."""
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))))[:, None]


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
