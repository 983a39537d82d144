"""The reference-compat surrogate of oracle/gp_compat.py against a LITERAL restatement of the reference's
kernel definitions (models/GP.py:28-179): the same lambdas, with jax.grad replaced by central finite
differences.  This pins the closed forms (shifted geometries, index convention, signs)."""
import numpy as np
import pytest

from oracle.equation import GradDependentNonlinear, sample_points
from oracle.gp_compat import OracleGPCompat, f16

D = 7
IDX = np.array([5, 0, 3, 6, 2])
H = 1e-3


def _gp():
    return OracleGPCompat(GradDependentNonlinear(D + 1), IDX, round16=False)


class Literal:
    """models/GP.py:28-179 with autodiff -> central differences (float64, no float16 casts)."""

    def __init__(self, gp):
        self.s2, self.d, self.idx = gp.s2, gp.d, gp.idx

    def kappa(self, x, y):
        return np.exp(-np.sum((x - y) ** 2) / (2 * self.s2))                     # :41-43

    @staticmethod
    def grad(f, v):
        out = np.zeros_like(v)
        for i in range(v.size):
            e = np.zeros_like(v)
            e[i] = H
            out[i] = (f(v + e) - f(v - e)) / (2 * H)
        return out

    def laplacian_op(self, f):                                                    # :28-39
        def lap(x):
            tot = 0.0
            for i in self.idx:
                e = np.zeros_like(x)
                e[i] = H
                tot += (f(x + e) - 2 * f(x) + f(x - e)) / (H * H)
            return tot / len(self.idx) * self.d
        return lap

    def dt_x(self, x, y):
        return self.grad(lambda v: self.kappa(v, y), x)[-1]                       # :59-63

    def dt_y(self, x, y):
        return self.grad(lambda v: self.kappa(x, v), y)[-1]                       # :69-73

    def div_x(self, x, y):
        return self.grad(lambda v: self.kappa(v, y), x)[:-1].sum()                # :75-79

    def div_y(self, x, y):
        return self.grad(lambda v: self.kappa(x, v), y)[:-1].sum()                # :81-85

    def lap_x_of(self, fn, x, y):                                                 # the pattern of :87-95, 151-179
        t_x, xs = x[0:1], x[1:]
        return self.laplacian_op(lambda v: fn(np.concatenate((v, t_x)), y))(xs)

    def lap_y_of(self, fn, x, y):                                                 # the pattern of :97-105, 119-149
        t_y, ys = y[0:1], y[1:]
        return self.laplacian_op(lambda v: fn(x, np.concatenate((v, t_y))))(ys)

    def block(self, opx, opy, x, y):
        base = {"I": self.kappa, "dt": None, "div": None}
        if opx == "lap" and opy == "lap":                                         # :171-179
            return self.lap_x_of(lambda a, b: self.lap_y_of(self.kappa, a, b), x, y)
        if opy == "lap":
            fn = {"I": self.kappa, "dt": self.dt_x, "div": self.div_x}[opx]       # :97-105, 119-127, 141-149
            return self.lap_y_of(fn, x, y)
        if opx == "lap":
            fn = {"I": self.kappa, "dt": self.dt_y, "div": self.div_y}[opy]       # :87-95, 151-169
            return self.lap_x_of(fn, x, y)
        raise KeyError((opx, opy, base))


@pytest.mark.parametrize("opx,opy", [("I", "lap"), ("lap", "I"), ("dt", "lap"), ("lap", "dt"), ("div", "lap"),
                                     ("lap", "div"), ("lap", "lap")])
def test_shifted_hutchinson_blocks_match_the_literal_definitions(opx, opy):
    gp = _gp()
    lit = Literal(gp)
    X, Y = sample_points(np.random.default_rng(3), D, 4, 3)
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    got = gp.block(opx, opy, X, Y)
    want = np.array([[lit.block(opx, opy, x, y) for y in Y] for x in X])
    tol = 2e-4 if (opx, opy) == ("lap", "lap") else 2e-5
    assert np.allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())


def test_compat_gram_is_symmetric():
    """Every block is a derivative of one scalar function of (x, y) (shifted or not), so K(phi, phi) stays symmetric --
    which is what lets the SVD factor U sqrt(S + nugget) be restated through an eigendecomposition (|K| + nugget I)."""
    gp = _gp()
    X, Y = sample_points(np.random.default_rng(4), D, 5, 5)
    K = gp.kernel_phi_phi(X, Y)
    assert np.allclose(K, K.T, atol=1e-12)


def test_float16_rounding_points():
    gp = OracleGPCompat(GradDependentNonlinear(D + 1), IDX, round16=True)
    X, Y = sample_points(np.random.default_rng(5), D, 6, 4)
    for ops in [("I", "I"), ("dt", "div"), ("lap", "lap"), ("I", "lap")]:
        B = gp.block(ops[0], ops[1], X, Y)
        assert np.array_equal(B, f16(B))
    gp.GPsolver(X, Y, GN_steps=3)
    assert np.array_equal(gp.cholesky_phi_phi_perturb, f16(gp.cholesky_phi_phi_perturb))
    assert np.isfinite(gp.right_vector).all()


def test_factor_rounding_is_immaterial_and_K_stays_positive_definite():
    """The product factors K + nugget I by Cholesky; the reference's float16 rounding of its SVD factor (models/GP.py:266)
    has no counterpart there.  Its effect on the surrogate is far below the solver tolerance, and the shifted blocks
    leave K positive definite at the reference's point densities, so |K| + nugget I = K + nugget I."""
    d, idx = 20, [11, 17, 12, 6, 4]
    dom, bdy = sample_points(np.random.default_rng(7), d, 160, 40)
    dom, bdy = f16(dom), f16(bdy)
    X = np.concatenate(sample_points(np.random.default_rng(8), d, 100, 30))
    preds = []
    for rf in (True, False):
        gp = OracleGPCompat(GradDependentNonlinear(d + 1), idx, round16=True, round_factor=rf, round_out=False)   # compare before the float16 cast of predict
        gp.GPsolver(dom, bdy, GN_steps=20)
        assert gp.K_eig_min > 0
        preds.append(gp.predict(X))
    assert np.abs(preds[0] - preds[1]).max() < 2e-4 * np.abs(preds[1]).max()


def test_compat_gradient_is_the_derivative_of_the_as_coded_posterior_mean():
    """compute_gradient (models/GP.py:673-687) is autodiff of dot(kernel_x_t_phi_single(x), right_vector): the oracle's closed form
    against central differences of its own (unrounded) predict, and its spatial sum against the as-coded div row."""
    rng = np.random.default_rng(21)
    gp = _gp()
    dom, bdy = sample_points(rng, D, 30, 9)
    gp.x_t_domain, gp.x_t_boundary = dom.astype(np.float64), bdy.astype(np.float64)
    gp.N_domain, gp.N_boundary, gp.phi_dim = 30, 9, 129
    gp.right_vector = rng.normal(size=(129, 1))
    X = rng.uniform(-0.5, 0.5, (11, D + 1))
    got = gp.compute_gradient(X)
    fd = np.zeros_like(got)
    for i in range(D + 1):
        e = np.zeros(D + 1)
        e[i] = 1e-5
        fd[:, i] = ((gp.predict(X + e) - gp.predict(X - e)) / 2e-5)[:, 0]
    assert np.abs(got - fd).max() <= 1e-7 * max(1.0, np.abs(fd).max())
    assert np.allclose(got[:, :D].sum(1), gp.div_x(X)[:, 0], rtol=1e-10, atol=1e-12)
    assert np.allclose(got[:, D], gp.pde_parts(X)[0][:, 0], rtol=1e-10, atol=1e-12)
    # with the float16 casts on, the result is a float16 value per coordinate (:687)
    g16 = OracleGPCompat(GradDependentNonlinear(D + 1), IDX, round16=True)
    g16.__dict__.update({k: v for k, v in gp.__dict__.items() if k not in ("round16", "round_out", "round_factor")})
    r = g16.compute_gradient(X)
    assert np.array_equal(r, f16(r)) and np.abs(r - got).max() <= 2.0 ** -11 * np.abs(got).max()
