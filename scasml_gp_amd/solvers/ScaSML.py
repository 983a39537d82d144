"""``ScaSML`` (solvers/ScaSML.py:5-304): multilevel Picard on the defect u - u_GP, on libscasml_hip."""
from .. import tables
from ._picard import PicardEngine, deliver


class ScaSML:
    '''Multilevel Picard Iteration calibrated GP for high dimensional semilinear PDE'''
    _variant = "quad"

    def __init__(self, equation, GP, seed=0, compat_crn=False, compat_f16=False, compat_rng=None, reference_mode=False):
        """reference_mode=True: the reference's random stream under its key schedule (compat_rng="jax") and its solver-level float16
        casts (compat_f16) in one switch; with the default GP (compat="reference") that is the reference's ScaSML."""
        self.equation = equation
        self.sigma = equation.sigma
        self.mu = equation.mu
        equation.geometry()
        self.T = equation.T
        self.t0 = equation.t0
        self.n_input = equation.n_input
        self.n_output = equation.n_output
        self.GP = GP
        self.evaluation_counter = 0
        self.key = seed
        self._engine = PicardEngine(equation, self._variant, gp=GP, seed=seed, compat_crn=compat_crn, compat_f16=compat_f16, compat_rng=compat_rng,
                                    reference_mode=reference_mode)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name == "GP" and "_engine" in self.__dict__:            # harness reassigns solver.GP
            self._engine.gp = value

    def f(self, x_t, u_breve, z_breve):
        '''solvers/ScaSML.py:29-47 (host view; the kernels fuse this into the tree walk).'''
        eq = self.equation
        u_hat = self.GP.predict(x_t)
        grad_x = self.GP.compute_gradient(x_t, u_hat)[:, :-1]
        return eq.f(x_t, u_breve + u_hat, eq.sigma(x_t) * grad_x + z_breve) - eq.f(x_t, u_hat, eq.sigma(x_t) * grad_x)

    def g(self, x_t):
        return (self.equation.g(x_t) - self.GP.predict(x_t))[:, 0]   # :49-63

    def inverse_gamma(self, gamma_input):
        '''Inverse of the gamma function through Lambert W (solvers/ScaSML.py:65-77); scalar or array in, the same out.'''
        return tables.inverse_gamma(gamma_input)

    def lgwt(self, N, a, b):
        '''(nodes, weights) exactly as the reference's routine returns them, its scalar quirk at :99 included (solvers/ScaSML.py:79-117).'''
        return tables.lgwt_reference(int(N), a, b)

    def approx_parameters(self, rhomax):
        return tables.approx_parameters(int(rhomax), float(self.T))

    def _solve(self, n, par, x_t):
        uz, uhat, was_numpy = self._engine.solve(int(n), int(par), x_t)
        self.evaluation_counter += self._engine.evaluation_increment(int(n), int(par))
        return uz, uhat, was_numpy

    def uz_solve(self, n, rho, x_t):
        '''solvers/ScaSML.py:149-284.'''
        self.Mf, self.Mg, self.Q, self.c, self.w = self.approx_parameters(rho)       # :161
        uz, _, was_numpy = self._solve(n, rho, x_t)
        return deliver(uz, was_numpy)

    def u_solve(self, n, rho, x_t):
        '''u_hat + u_breve, solvers/ScaSML.py:286-304.'''
        self.Mf, self.Mg, self.Q, self.c, self.w = self.approx_parameters(rho)
        uz, uhat, was_numpy = self._solve(n, rho, x_t)
        return deliver(self._sum16(uz[:, 0:1] + uhat[:, None]), was_numpy)

    def _sum16(self, total):
        """compat_f16: u_breve + u_hat is a float16 sum in the reference (:300-304: u_breve float16 by :284 -- in the full-history solver by
        float16 arithmetic throughout -- and u_hat float16 by models/GP.py:671); held in float32."""
        import torch
        return total.to(torch.float16).to(torch.float32) if self._engine.compat_f16 else total
