"""Deterministic pins of the oracle: integer tables, reference-compat lgwt rows
(SURVEY.md Appendix B), recursion work counts (the call counts
19 and 5 are what the reference's committed cProfile dumps record), exact solution."""
import json
import os

import numpy as np
import pytest

from oracle.equation import GradDependentNonlinear
from oracle.mlp import reference_counts, site_count
from oracle.tables import approx_parameters, inverse_gamma, lgwt

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tables_appendix_b.json")))


@pytest.mark.parametrize("rho", [1, 2, 3, 4, 5])
def test_integer_tables(rho):
    Mf, Mg, Q, c, w = approx_parameters(rho)
    assert list(Q[rho - 1, :rho]) == GOLD["Q"][str(rho)]
    assert list(Mf[rho - 1, :rho]) == GOLD["Mf"][str(rho)]
    assert list(Mg[rho - 1, :rho + 1]) == GOLD["Mg"][str(rho)]


def test_inverse_gamma_is_far_from_rounding_boundaries():
    for x, v in ((3 ** 1.5, 3.880), (3.0, 3.404), (2.0, 3.002)):
        assert abs(inverse_gamma(x) - v) < 2e-3


@pytest.mark.parametrize("q", [1, 3, 4])
def test_lgwt_reference_compat_rows(q):
    x, w = lgwt(q, 0, 0.5)
    assert np.allclose(x, GOLD["lgwt"][str(q)]["nodes"], rtol=0, atol=1e-15)
    assert np.allclose(w, GOLD["lgwt"][str(q)]["weights"], rtol=0, atol=1e-15)


def test_lgwt_two_nodes_has_nan_weight():
    x, w = lgwt(2, 0, 0.5)
    assert abs(x[0] - 0.39433756729740643) < 1e-15 and x[1] == 0 and np.isnan(w[1])


@pytest.mark.parametrize("key", sorted(GOLD["counts"].keys()))
def test_reference_work_counts(key):
    variant, n, par = key.split(",")
    n, par = int(n), int(par)
    tab = approx_parameters(par) if variant == "quad" else None
    got = reference_counts(variant, n, par, tab, scasml=True)
    exp = GOLD["counts"][key]
    for k in ("calls", "jumps", "steps", "f", "pde", "path_steps"):
        assert got[k] == exp[k], (key, k)
    assert site_count(variant, n, par, tab) == exp["executed"]


def test_exact_solution_formula():
    eq = GradDependentNonlinear(4)
    x = np.array([[0.1, -0.2, 0.3, 0.25]])
    assert np.isclose(eq.exact_solution(x)[0, 0], 1 - 1 / (1 + np.exp(0.45)))
    assert np.isclose(eq.mu(), -1 / 3 - 0.25 ** 2 / 2) and eq.sigma() == 0.25
    assert np.allclose(eq.g(x), eq.exact_solution(x))


def test_solver_classes_expose_inverse_gamma_and_lgwt():
    """solvers/MLP.py:57-109 (and the copy in ScaSML.py:65-117): the two table routines are methods of the reference's solver classes; the
    drop-in classes keep them (host NumPy: no GPU needed to construct an MLP).  Values: SURVEY.md Appendix B."""
    import numpy as np
    from oracle import tables as otab
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    s = MLP(Grad_Dependent_Nonlinear(21))
    assert abs(s.inverse_gamma(3.0 ** 1.5) - 3.880) < 1e-3 and abs(s.inverse_gamma(2.0) - 3.002) < 1e-3
    assert np.allclose(s.inverse_gamma(np.array([2.0, 3.0])), [otab.inverse_gamma(2.0), otab.inverse_gamma(3.0)], rtol=1e-15)
    x, w = s.lgwt(3, 0.0, 0.5)
    xo, wo = otab.lgwt(3, 0.0, 0.5)
    assert np.array_equal(x, xo) and np.array_equal(w, wo) and abs(w.sum() - 0.2598) < 1e-4           # not Gauss-Legendre (E-1): the reference's rule
    x1, w1 = s.lgwt(1, 0.0, 0.5)
    assert np.allclose(x1, [0.25]) and np.allclose(w1, [0.5])
    assert np.isnan(s.lgwt(2, 0.0, 0.5)[1]).any()                                                       # the q = 2 NaN weight
    assert ScaSML.inverse_gamma is not None and ScaSML.lgwt is not None
