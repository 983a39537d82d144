"""``MLP_full_history`` (solvers/MLP_full_history.py:6-196) on libscasml_hip."""
from ._picard import PicardEngine, deliver


class MLP_full_history:
    '''Full-history multilevel Picard: one uniformly random time per sample, M^n / M^(n-l) samples.'''
    _variant = "fh"

    def __init__(self, equation, seed=0, compat_crn=False, compat_f16=False, compat_rng=None, reference_mode=False):
        """reference_mode=True: the reference's random stream (every draw from the one key of MLP_full_history.py:92-93) and its
        solver-level float16 casts, in one switch."""
        self.equation = equation
        self.sigma = equation.sigma
        self.mu = equation.mu
        equation.geometry()
        self.T = equation.T
        self.t0 = equation.t0
        self.n_input = equation.n_input
        self.n_output = equation.n_output
        self.evaluation_counter = 0
        self._engine = PicardEngine(equation, self._variant, gp=None, seed=seed, compat_crn=compat_crn, compat_f16=compat_f16, compat_rng=compat_rng,
                                    reference_mode=reference_mode)

    def f(self, x_t, u, z):
        return self.equation.f(x_t, u, z)

    def g(self, x_t):
        return self.equation.g(x_t)[:, 0]

    def uz_solve(self, n, rho, x_t, M):
        '''solvers/MLP_full_history.py:64-180 (rho is ignored there as well).'''
        uz, _, was_numpy = self._engine.solve(int(n), int(M), x_t)
        self.evaluation_counter += self._engine.evaluation_increment(int(n), int(M))
        return deliver(uz, was_numpy)

    def u_solve(self, n, rho, x_t, M=3):
        return self.uz_solve(n, rho, x_t, M)[:, 0:1]              # :182-196
