"""``ScaSML_full_history`` (solvers/ScaSML_full_history.py:5-221) on libscasml_hip."""
from .ScaSML import ScaSML
from ._picard import deliver


class ScaSML_full_history(ScaSML):
    '''Full-history multilevel Picard on the defect u - u_GP.'''
    _variant = "fh"

    def uz_solve(self, n, rho, x_t, M):
        '''solvers/ScaSML_full_history.py:75-199.'''
        uz, _, was_numpy = self._solve(n, M, x_t)
        return deliver(uz, was_numpy)

    def u_solve(self, n, rho, x_t, M=3):
        uz, uhat, was_numpy = self._solve(n, M, x_t)               # :201-221
        return deliver(self._sum16(uz[:, 0:1] + uhat[:, None]), was_numpy)
