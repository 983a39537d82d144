"""Problem definitions with the reference's call surface (equations/equations.py).

``Equation`` (:15-230) is the abstract base the solvers and the GP take.  The kernels cover the semilinear family
    u_t + mu sum_i d_i u + sigma^2/2 Lap u + f(u, sum_i z_i) = 0,  z = sigma grad u,  u(T) = g,   mu, sigma constant,
through an equation registry (``eq_id``; device functors in csrc/equations.hpp): the reference's concrete
``Grad_Dependent_Nonlinear`` (:232-417) is id 0 and ``Cubic_Reaction_Diffusion`` (no reference counterpart) id 1.  The
methods here are the host-side view of the same formulas (NumPy, for the harness metric and the GP boundary data); a
subclass that sets no ``eq_id`` has no kernels and the solvers refuse it.  deepxde is
not a dependency: ``geometry()`` records the box and the samplers are NumPy restatements of
``GeometryXTime.random_points / random_boundary_points``.
"""
import numpy as np

from .. import _lib


class _Box:
    """Stand-in for the deepxde geometry objects the reference stores in geomx / geomt."""

    def __init__(self, lo, hi):
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        self.dim = self.lo.size


class Equation(object):
    """Abstract PDE  u_t + mu.grad u + sigma^2/2 Lap u + f(u, sigma grad u) = 0, u(T) = g
    (equations/equations.py:15-230)."""
    eq_id = None           # kernels exist only for subclasses that set this

    def __init__(self, n_input, n_output=1):
        self.n_input = n_input
        self.n_output = n_output

    def f(self, x_t, u, z):
        raise NotImplementedError

    def g(self, x_t):
        return self.terminal_constraint(x_t)          # equations.py:146-162

    def terminal_constraint(self, x_t):
        raise NotImplementedError

    def mu(self, x_t=0):
        raise NotImplementedError

    def sigma(self, x_t=0):
        raise NotImplementedError

    def exact_solution(self, x_t):
        raise NotImplementedError

    def geometry(self, t0=0, T=0.5):
        raise NotImplementedError

    def generate_data(self, num_domain=100, num_boundary=20):
        raise NotImplementedError

    def generate_test_data(self, num_domain=100, num_boundary=20):
        raise NotImplementedError


class _LogisticWave(Equation):
    """Shared by the registered equations: unit-cube geometry, terminal condition and closed-form solution
    1 - 1/(1 + exp(t + sum x)) (equations/equations.py:248-261, 306-323, 344-417), sigma = 0.25."""

    def __init__(self, n_input, n_output=1):
        super().__init__(n_input, n_output)
        self.uncertainty = 1e-1          # :245
        self.norm_estimation = 1         # :246

    def terminal_constraint(self, x_t):
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :self.n_input - 1], axis=1))))[:, None]   # :259

    def sigma(self, x_t=0):
        return 0.25                                        # :288

    def f(self, x_t, u, z):
        return self.f_parts(np.asarray(u, dtype=np.float64), np.sum(np.asarray(z, dtype=np.float64), axis=1, keepdims=True))[0]

    def F_parts(self, z1, z3, z5):
        """The GP's collocation operator u_t = F(u, Lap u, div u) and its derivatives (models/GP.py:705-743), host view of
        csrc/equations.hpp eq_F: -> F, (dF/dz1, dF/dz3, dF/dz5)."""
        s, mu = self.sigma(), self.mu()
        f, fu, fs = self.f_parts(np.asarray(z1, dtype=np.float64), s * np.asarray(z5, dtype=np.float64))[:3]
        return -mu * z5 - (s ** 2 / 2) * z3 - f, (-fu, -(s ** 2 / 2) * np.ones_like(z1), -mu - s * fs)

    def exact_solution(self, x_t):
        x_t = np.asarray(x_t, dtype=np.float64)
        e = np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))       # :317-321
        return (1 - 1 / (1 + e))[:, None]

    def exact_solution_derivative(self, x_t):
        x_t = np.asarray(x_t, dtype=np.float64)
        e = np.exp(x_t[:, -1] + np.sum(x_t[:, :-1], axis=1))       # :336-340
        return (e / (1 + e) ** 2)[:, None]

    def geometry(self, t0=0, T=0.5):
        self.t0, self.T, self.radius = t0, T, 0.5                   # :355-357
        d = self.n_input - 1
        self.geomx = _Box([-self.radius] * d, [self.radius] * d)
        self.geomt = _Box([t0], [T])
        return self

    def test_geometry(self, t0=0, T=0.5):
        self.t0, self.T, self.test_T, self.test_radius = t0, T, T, 0.5   # :376-379
        d = self.n_input - 1
        self.geomx = _Box([-self.test_radius] * d, [self.test_radius] * d)
        self.geomt = _Box([t0], [T])
        return self

    def _sample(self, num_domain, num_boundary):
        """Uniform interior points; boundary points = uniform in the cube with one random
        coordinate snapped to a face, at a uniform random time (NumPy global RNG, as deepxde's
        "pseudo" sampler; cast to float16 like dde's default float, experiment_run.py:30)."""
        d = self.n_input - 1
        lo, hi = self.geomx.lo, self.geomx.hi
        dom = np.concatenate([np.random.random((num_domain, d)) * (hi - lo) + lo,
                              np.random.permutation(np.random.random((num_domain, 1)) * (self.T - self.t0) + self.t0)], axis=1)
        xb = np.random.random((num_boundary, d))
        pick = np.random.randint(d, size=num_boundary)
        xb[np.arange(num_boundary), pick] = np.round(xb[np.arange(num_boundary), pick])
        bdy = np.concatenate([xb * (hi - lo) + lo,
                              np.random.permutation(np.random.random((num_boundary, 1)) * (self.T - self.t0) + self.t0)], axis=1)
        return dom.astype(np.float16), bdy.astype(np.float16)

    def generate_data(self, num_domain=100, num_boundary=20):
        self.geometry()                                             # :398-401
        return self._sample(num_domain, num_boundary)

    def generate_test_data(self, num_domain=100, num_boundary=20):
        self.test_geometry()                                        # :414-417
        return self._sample(num_domain, num_boundary)


class Grad_Dependent_Nonlinear(_LogisticWave):
    """equations/equations.py:232-417: f = sigma u sum_i z_i, mu = -1/d - sigma^2/2."""
    eq_id = _lib.EQ_GRAD_DEPENDENT_NONLINEAR

    def mu(self, x_t=0):
        s = self.sigma()
        return -1 / (self.n_input - 1) - s ** 2 / 2        # :273-276

    def f_parts(self, u, sz):
        """(f, df/du, df/ds, ...) of f(u, s) = sigma u s (:303), s = sum_i z_i."""
        s = self.sigma()
        z = np.zeros_like(u)
        return s * u * sz, s * sz, s * u, z, s + z, z


class Cubic_Reaction_Diffusion(_LogisticWave):
    """A second registered equation (no reference counterpart: the reference ships one PDE; this one exercises the generic
    ``Equation`` path): u_t + sigma^2/2 Lap u - u (1 - u) (1 + c (1 - 2u)) = 0 with c = sigma^2 d / 2, mu = 0, whose travelling
    wave 1 - 1/(1 + exp(t + sum x)) is an exact solution.  f does not depend on the gradient."""
    eq_id = _lib.EQ_CUBIC_REACTION_DIFFUSION

    def mu(self, x_t=0):
        return 0.0

    def f_parts(self, u, sz):
        c = self.sigma() ** 2 * (self.n_input - 1) / 2
        w, v = u * (1 - u), 1 + c * (1 - 2 * u)
        z = np.zeros_like(u)
        return -w * v, -(1 - 2 * u) * v + 2 * c * w, z, 2 * v + 4 * c * (1 - 2 * u), z, z
