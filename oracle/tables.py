"""Oracle quadrature / sample-count tables.  TEST INFRASTRUCTURE (oracle/__init__.py).

Literal float64 restatement of solvers/MLP.py:57-139 (duplicated at
solvers/ScaSML.py:65-147), including the scalar assignment at MLP.py:99 that makes
``lgwt`` differ from Gauss-Legendre for q >= 2.
The reference runs these with ``jax_enable_x64`` (experiment_run.py:46), i.e. IEEE
float64, which NumPy reproduces op for op.
"""
import numpy as np
from scipy.special import lambertw


def inverse_gamma(gamma_input):
    """solvers/MLP.py:57-69."""
    c = 0.036534
    L = np.log((gamma_input + c) / np.sqrt(2 * np.pi))
    return float(np.real(L / np.real(lambertw(L / np.e)) + 0.5))


def lgwt(N, a, b):
    """solvers/MLP.py:71-109, shapes and operation order kept as written."""
    N -= 1
    N1, N2 = N + 1, N + 2
    xu = np.linspace(-1, 1, N1).reshape(1, -1)
    y = np.cos((2 * np.arange(0, N + 1, 1) + 1) * np.pi / (2 * N + 2)) + (0.27 / N1) * np.sin(np.pi * xu * N / N2)
    L = np.zeros((N1, N2))
    Lp = np.zeros((N1, N2))
    y0 = 2
    eps = 2.2204e-16
    iteration = 0
    max_iter = 100
    with np.errstate(all="ignore"):
        while np.max(np.abs(y - y0)) > eps and iteration < max_iter:
            L[:, 0] = 1
            L[:, 1] = y[0, 0]                      # MLP.py:99 -- scalar, not the vector y
            for k in range(2, N1 + 1):
                L[:, k] = (((2 * k - 1) * y * L[:, k - 1] - (k - 1) * L[:, k - 2]) / k)[0]
            Lp = (N2) * (L[:, N1 - 1] - y * L[:, N2 - 1]) / (1 - y * y)
            y0 = y
            y = y0 - L[:, N2 - 1] / Lp
            iteration += 1
        x = (a * (1 - y) + b * (1 + y)) / 2
        w = (b - a) / ((1 - y * y) * (Lp * Lp)) * (N2 * N2) / (N1 * N1)
    return x[0], w[0]


def approx_parameters(rhomax, T=0.5):
    """solvers/MLP.py:111-139 -> (Mf, Mg, Q, c, w)."""
    Q = np.zeros((rhomax, rhomax), dtype=np.int64)
    Mf = np.zeros((rhomax, rhomax), dtype=np.int64)
    Mg = np.zeros((rhomax, rhomax + 1), dtype=np.int64)
    for rho in range(1, rhomax + 1):
        for k in range(1, rho + 1):
            Q[rho - 1, k - 1] = int(np.round(inverse_gamma(rho ** (k / 2))))
            Mf[rho - 1, k - 1] = int(np.round(rho ** (k / 2)))
            Mg[rho - 1, k - 1] = int(np.round(rho ** (k - 1)))
        Mg[rho - 1, rho] = rho ** rho
    qmax = int(np.max(Q))
    c = np.zeros((qmax, qmax))
    w = np.zeros((qmax, qmax))
    for k in range(1, qmax + 1):
        ctemp, wtemp = lgwt(k, 0, T)
        c[:, k - 1] = np.concatenate([ctemp[::-1], np.zeros(qmax - k)])
        w[:, k - 1] = np.concatenate([wtemp[::-1], np.zeros(qmax - k)])
    return Mf, Mg, Q, c, w
