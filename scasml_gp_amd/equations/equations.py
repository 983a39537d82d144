"""Problem definitions with the reference's call surface (equations/equations.py).

``Equation`` (:15-230) is the abstract base the solvers and the GP take; only the concrete
``Grad_Dependent_Nonlinear`` (:232-417) is on the hot path.  Its f / g / mu / sigma are what the
HIP kernels hard-wire under ``eq_id`` (include/scasml_hip.h); the methods here are the host-side
view of the same formulas (NumPy, for the harness metric and the GP boundary data).  deepxde is
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


class Grad_Dependent_Nonlinear(Equation):
    """equations/equations.py:232-417."""
    eq_id = _lib.EQ_GRAD_DEPENDENT_NONLINEAR

    def __init__(self, n_input, n_output=1):
        super().__init__(n_input, n_output)
        self.uncertainty = 1e-1          # :245
        self.norm_estimation = 1         # :246

    def terminal_constraint(self, x_t):
        x_t = np.asarray(x_t, dtype=np.float64)
        return (1 - 1 / (1 + np.exp(x_t[:, -1] + np.sum(x_t[:, :self.n_input - 1], axis=1))))[:, None]   # :259

    def mu(self, x_t=0):
        s = self.sigma()
        return -1 / (self.n_input - 1) - s ** 2 / 2        # :273-276

    def sigma(self, x_t=0):
        return 0.25                                        # :288

    def f(self, x_t, u, z):
        return self.sigma() * np.asarray(u, dtype=np.float64) * np.sum(np.asarray(z, dtype=np.float64), axis=1, keepdims=True)  # :303

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
