"""The semilinear family of oracle/equation.py: the reference's Grad_Dependent_Nonlinear and the second registered
equation (Cubic_Reaction_Diffusion).  For both: the closed-form solution satisfies the PDE, f_parts / F_parts are consistent
with finite differences, and the multilevel-Picard oracle converges towards the exact solution."""
import numpy as np
import pytest

from oracle.equation import CubicReactionDiffusion, GradDependentNonlinear, rel_l2, sample_points
from oracle.mlp import PicardOracle

EQS = [GradDependentNonlinear, CubicReactionDiffusion]


@pytest.mark.parametrize("cls", EQS)
def test_exact_solution_satisfies_the_pde(cls):
    d, h = 6, 1e-4
    eq = cls(d + 1)
    P = np.concatenate(sample_points(np.random.default_rng(0), d, 20, 0)).astype(np.float64)
    u = eq.exact_solution(P)[:, 0]

    def shifted(k, delta):
        Q = P.copy()
        Q[:, k] += delta
        return eq.exact_solution(Q)[:, 0]
    ut = (shifted(d, h) - shifted(d, -h)) / (2 * h)
    grad = np.stack([(shifted(k, h) - shifted(k, -h)) / (2 * h) for k in range(d)], axis=1)
    lap = sum((shifted(k, h) - 2 * u + shifted(k, -h)) / (h * h) for k in range(d))
    s = eq.sigma()
    res = ut + eq.mu() * grad.sum(1) + s * s / 2 * lap + eq.f(P, u[:, None], s * grad)[:, 0]
    assert np.abs(res).max() < 1e-6
    assert np.allclose(eq.g(np.concatenate([P[:, :d], np.full((len(P), 1), eq.T)], axis=1)),
                       eq.exact_solution(np.concatenate([P[:, :d], np.full((len(P), 1), eq.T)], axis=1)))


@pytest.mark.parametrize("cls", EQS)
def test_derivatives_of_f_and_F(cls):
    eq = cls(11)
    rng = np.random.default_rng(1)
    u, s = rng.uniform(0.1, 0.9, 50), rng.uniform(-1, 1, 50)
    h = 1e-5
    f, fu, fs, fuu, fus, fss = eq.f_parts(u, s)
    assert np.allclose(fu, (eq.f_parts(u + h, s)[0] - eq.f_parts(u - h, s)[0]) / (2 * h), atol=1e-8)
    assert np.allclose(fs, (eq.f_parts(u, s + h)[0] - eq.f_parts(u, s - h)[0]) / (2 * h), atol=1e-8)
    assert np.allclose(fuu, (eq.f_parts(u + h, s)[1] - eq.f_parts(u - h, s)[1]) / (2 * h), atol=1e-7)
    assert np.allclose(fus, (eq.f_parts(u, s + h)[1] - eq.f_parts(u, s - h)[1]) / (2 * h), atol=1e-7)
    assert np.allclose(fss, (eq.f_parts(u, s + h)[2] - eq.f_parts(u, s - h)[2]) / (2 * h), atol=1e-7)
    z1, z3, z5 = u, rng.standard_normal(50), s
    F, (d1, d3, d5), (F11, F15, F55) = eq.F_parts(z1, z3, z5)
    assert np.allclose(d1, (eq.F_parts(z1 + h, z3, z5)[0] - eq.F_parts(z1 - h, z3, z5)[0]) / (2 * h), atol=1e-8)
    assert np.allclose(d3, (eq.F_parts(z1, z3 + h, z5)[0] - eq.F_parts(z1, z3 - h, z5)[0]) / (2 * h), atol=1e-8)
    assert np.allclose(d5, (eq.F_parts(z1, z3, z5 + h)[0] - eq.F_parts(z1, z3, z5 - h)[0]) / (2 * h), atol=1e-8)
    assert np.allclose(F15, (eq.F_parts(z1, z3, z5 + h)[1][0] - eq.F_parts(z1, z3, z5 - h)[1][0]) / (2 * h), atol=1e-7)
    assert np.allclose(F11, (eq.F_parts(z1 + h, z3, z5)[1][0] - eq.F_parts(z1 - h, z3, z5)[1][0]) / (2 * h), atol=1e-7)
    assert np.allclose(F55, (eq.F_parts(z1, z3, z5 + h)[1][2] - eq.F_parts(z1, z3, z5 - h)[1][2]) / (2 * h), atol=1e-7)


def test_grad_dependent_nonlinear_keeps_the_reference_forms():
    eq = GradDependentNonlinear(21)
    z1, z3, z5 = np.array([0.3]), np.array([-0.2]), np.array([0.7])
    s, d = 0.25, 20
    assert np.allclose(eq.F_parts(z1, z3, z5)[0], -s ** 2 * z1 * z5 + (1 / d + s ** 2 / 2) * z5 - (s ** 2 / 2) * z3)   # models/GP.py:717
    assert np.allclose(eq.f(None, np.array([[0.3]]), np.full((1, d), 0.1)), s * 0.3 * 2.0)                          # equations.py:303


def test_picard_oracle_converges_on_the_second_equation():
    d = 10
    eq = CubicReactionDiffusion(d + 1)
    xt = np.concatenate(sample_points(np.random.default_rng(2), d, 300, 60))
    exact = eq.exact_solution(xt)
    e2 = rel_l2(PicardOracle(eq, "quad", seed=1).u_solve(2, 2, xt), exact)
    e3 = rel_l2(PicardOracle(eq, "quad", seed=1).u_solve(3, 3, xt), exact)
    assert e3 < e2 < 0.25 and e3 < 0.12
