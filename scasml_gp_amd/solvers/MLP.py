"""``MLP`` with the reference's call surface (solvers/MLP.py:5-288), running on libscasml_hip."""
from .. import tables
from ._picard import PicardEngine, deliver


class MLP:
    '''Multilevel Picard Iteration for high dimensional semilinear PDE (solvers/MLP.py:5-25)'''
    _variant = "quad"

    def __init__(self, equation, seed=0, compat_crn=False, compat_f16=False, compat_rng=None, reference_mode=False):
        """reference_mode=True: what the reference's solver object computes -- its random stream under its key schedule
        (compat_rng="jax") and its solver-level float16 casts (compat_f16) -- in one switch."""
        self.equation = equation
        self.sigma = equation.sigma
        self.mu = equation.mu
        equation.geometry()
        self.T = equation.T
        self.t0 = equation.t0
        self.n_input = equation.n_input
        self.n_output = equation.n_output
        self.evaluation_counter = 0
        self.key = seed                      # Philox seed; replaces random.PRNGKey(0) (:25)
        self._engine = PicardEngine(equation, self._variant, gp=None, seed=seed, compat_crn=compat_crn, compat_f16=compat_f16, compat_rng=compat_rng,
                                    reference_mode=reference_mode)

    def f(self, x_t, u, z):
        return self.equation.f(x_t, u, z)                         # :27-41

    def g(self, x_t):
        return self.equation.g(x_t)[:, 0]                         # :43-55

    def inverse_gamma(self, gamma_input):
        '''Inverse of the gamma function through Lambert W (solvers/MLP.py:57-69); scalar or array in, the same out.'''
        return tables.inverse_gamma(gamma_input)

    def lgwt(self, N, a, b):
        '''(nodes, weights) exactly as the reference's routine returns them, its scalar quirk at :99 included (solvers/MLP.py:71-109).'''
        return tables.lgwt_reference(int(N), a, b)

    def approx_parameters(self, rhomax):
        return tables.approx_parameters(int(rhomax), float(self.T))   # :111-139 (cached)

    def uz_solve(self, n, rho, x_t):
        '''(batch, 1 + d): u and z of solvers/MLP.py:141-274.'''
        self.Mf, self.Mg, self.Q, self.c, self.w = self.approx_parameters(rho)
        uz, _, was_numpy = self._engine.solve(int(n), int(rho), x_t)
        self.evaluation_counter += self._engine.evaluation_increment(int(n), int(rho))
        return deliver(uz, was_numpy)

    def u_solve(self, n, rho, x_t):
        return self.uz_solve(n, rho, x_t)[:, 0:1]                 # :276-288
