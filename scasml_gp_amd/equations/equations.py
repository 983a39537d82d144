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

    # the abstract surface of equations/equations.py:31-230: every method a subclass may define; the ones the solvers and the
    # surrogate call (f, g / terminal_constraint, mu, sigma, exact_solution, geometry, generate_*) and the PINN-side ones the
    # reference declares but this path never calls (PDE_loss, gPDE_loss, the other constraints, data_loss)
    def PDE_loss(self, x_t, u, z):
        raise NotImplementedError                      # :31-43

    def gPDE_loss(self, x_t, u):
        raise NotImplementedError                      # :45-56

    def terminal_constraint(self, x_t):
        raise NotImplementedError                      # :58-68

    def initial_constraint(self, x_t):
        raise NotImplementedError                      # :70-80

    def Dirichlet_boundary_constraint(self, x_t):
        raise NotImplementedError                      # :82-92

    def Neumann_boundary_constraint(self, x_t):
        raise NotImplementedError                      # :94-104

    def f(self, x_t, u, z):
        raise NotImplementedError                      # :130-144

    def g(self, x_t):
        if hasattr(self, "terminal_constraint"):       # :146-162
            return self.terminal_constraint(x_t)
        raise NotImplementedError

    def data_loss(self, x_t):
        raise NotImplementedError                      # :176-186

    def mu(self, x_t=0):
        raise NotImplementedError

    def sigma(self, x_t=0):
        raise NotImplementedError

    def exact_solution(self, x_t):
        raise NotImplementedError

    def geometry(self, t0=0, T=0.5):
        raise NotImplementedError

    def test_geometry(self, t0=0, T=0.5):
        raise NotImplementedError                      # :201-212

    def generate_data(self, num_domain=100, num_boundary=20):
        raise NotImplementedError

    def generate_test_data(self, num_domain=100, num_boundary=20):
        raise NotImplementedError


def _logistic_wave_f16(x_t):
    """1 - 1/(1 + exp(t + sum x)) for FLOAT16 rows exactly as the reference's jitted float16 graph evaluates it
    (equations/equations.py:259, 317-322): jnp.sum accumulates float16 inputs in float32 and rounds once, every other operation is
    a float16 operation (computed exactly, rounded to float16).  Pinned by the "Real Solution" value the reference printed:
    ||exact||_2 / sqrt(n) of its test set agrees to all 16 digits at d = 20, 40, 60, 80 (tests/test_reference_logs.py)."""
    f16, f32 = np.float16, np.float32
    x = np.asarray(x_t, dtype=f16)
    s = x[:, :-1].astype(f32).sum(axis=1, dtype=f32).astype(f16)
    arg = (x[:, -1].astype(f32) + s.astype(f32)).astype(f16)
    with np.errstate(over="ignore"):                 # exp overflows float16 at t + sum x > 11.09 (inf, as in the reference): 1 - 1/(1 + inf) = 1
        e = np.exp(arg.astype(np.float64)).astype(f32).astype(f16)     # correctly rounded float32 exp, then float16 (machine-independent)
    q = (f32(1) / (f32(1) + e.astype(f32)).astype(f16).astype(f32)).astype(f16)
    return (f32(1) - q.astype(f32)).astype(f16)[:, None]


class _LogisticWave(Equation):
    """Shared by the registered equations: unit-cube geometry, terminal condition and closed-form solution
    1 - 1/(1 + exp(t + sum x)) (equations/equations.py:248-261, 306-323, 344-417), sigma = 0.25."""

    def __init__(self, n_input, n_output=1):
        super().__init__(n_input, n_output)
        self.uncertainty = 1e-1          # :245
        self.norm_estimation = 1         # :246

    def terminal_constraint(self, x_t):
        if np.asarray(x_t).dtype == np.float16:       # the reference's own arrays (deepxde float16): its float16 graph, float16 out (:259-261)
            return _logistic_wave_f16(x_t)
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
        if np.asarray(x_t).dtype == np.float16:       # :317-323 on float16 rows: float16 out, as the harness receives it
            return _logistic_wave_f16(x_t)
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
        """deepxde's samplers restated call for call, so that -- NumPy's global generator being the
        only source of randomness, ``sample(n, dim, "pseudo") = np.random.random((n, dim)).astype(float16)`` under
        ``dde.config.set_default_float("float16")`` (experiment_run.py:30) -- a given ``np.random.seed`` yields the reference's
        own points:
        * ``GeometryXTime.random_points`` of a Hypercube geometry samples the (d+1)-dimensional box [xmin, t0] .. [xmax, T] in
          ONE draw: ``(xmax - xmin) * sample(n, d + 1) + xmin``, float16 arithmetic;
        * ``GeometryXTime.random_boundary_points``: ``Hypercube.random_boundary_points`` (``x = sample(n, d)``,
          ``rand_dim = np.random.randint(d, size=n)``, that coordinate rounded to 0 or 1, then ``(xmax - xmin) * x + xmin``),
          and ``TimeDomain.random_points`` (``diam * sample(n, 1) + l``) permuted by ``np.random.permutation``."""
        d = self.n_input - 1
        f16 = np.float16
        lo = np.append(self.geomx.lo, self.t0).astype(f16)
        hi = np.append(self.geomx.hi, self.T).astype(f16)
        dom = (hi - lo) * np.random.random(size=(num_domain, d + 1)).astype(f16) + lo
        xb = np.random.random(size=(num_boundary, d)).astype(f16)
        pick = np.random.randint(d, size=num_boundary)
        xb[np.arange(num_boundary), pick] = np.round(xb[np.arange(num_boundary), pick])
        xb = (hi[:d] - lo[:d]) * xb + lo[:d]
        t = (f16(self.T - self.t0) * np.random.random(size=(num_boundary, 1)).astype(f16) + f16(self.t0)).astype(f16)
        t = np.random.permutation(t)
        bdy = np.hstack((xb, t))
        return dom.astype(f16), bdy.astype(f16)

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


class Quadratic_Gradient_Reaction_Diffusion(_LogisticWave):
    """A third registered equation, one step wider than the family above (no reference counterpart): f(u, sum_i z_i, sum_i z_i^2),
        u_t + sigma^2/2 Lap u - u (1 - u) (1 + c (1 - 2u)) + (|z|^2 - sigma^2 d (u (1 - u))^2) = 0,   c = sigma^2 d / 2,  z = sigma grad u,
    mu = 0; on the travelling wave z_i = sigma u (1 - u), so the gradient term vanishes and 1 - 1/(1 + exp(t + sum x)) stays the exact solution.
    Kernels: the surrogate-free Picard tree (MLP, MLP_full_history) -- one more xor-shuffle sum per evaluation of f.  ScaSML on it would need the
    surrogate's FULL gradient at every tree site (the fused evaluation delivers div u_hat): refused, as the GP fit is (its collocation operator is
    a function of (u, Lap u, div u) alone, models/GP.py:705-719)."""
    eq_id = _lib.EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION
    surrogate_free_only = True

    def mu(self, x_t=0):
        return 0.0

    def f(self, x_t, u, z):
        u, z = np.asarray(u, dtype=np.float64), np.asarray(z, dtype=np.float64)
        s, d = self.sigma(), self.n_input - 1
        w = u * (1 - u)
        return -w * (1 + (s * s * d / 2) * (1 - 2 * u)) + (np.sum(z * z, axis=1, keepdims=True) - s * s * d * w * w)

    def f_parts(self, u, sz):
        raise NotImplementedError("Quadratic_Gradient_Reaction_Diffusion: f depends on |z|^2, not on sum z alone: no (u, sum z) form (and no GP collocation operator)")

