"""Study (not a test): which reading of `-sum(...) / (2 sigma**2)` on float16 rows do the reference's logs support -- a float16 division by the
weakly typed constant, or the multiplication by the folded reciprocal that XLA's algebraic simplifier emits for a division by a constant?
GP relative L2 of oracle/gp_compat.py (f16_graph=True) against results/**/SimpleUniform.log:4 at d = 20, 40, 60, 80.  About three minutes.
    python tests/studies/f16_graph_study.py
(The study below is level 1 -- kappa and the first-order blocks; levels 2 and 3 of OracleGPCompat(f16_graph=...) were measured the same way:
 -2.9e-6 / +5.8e-6 / -1.5e-5 / -1.7e-5 and +1.3e-5 / -1.9e-5 / -3.2e-5 / -1.2e-5.)
Result (round 4): reciprocal +1.3e-5 / -1.6e-6 / -1.6e-5 / -2.5e-5; division -1.5e-5 / -6.2e-5 / +8.6e-5 / -1.0e-5; L1 max with the reciprocal
0.354980 / 0.400879 / 0.383301 / 0.361328 against the logged 0.354492 / 0.400391 / 0.383545 / 0.361328."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle.equation import GradDependentNonlinear, deepxde_points, logistic_wave_f16
from oracle.gp_compat import OracleGPCompat
from scasml_gp_amd.threefry import reference_laplacian_idx
F16, F32 = np.float16, np.float32
LOG = json.load(open('tests/golden/reference_logged.json'))

def make(variant):
    class G(OracleGPCompat):
        def _f16_first_order(self, opx, opy, X, Y):
            d = self.d
            X16, Y16 = np.asarray(X).astype(F16), np.asarray(Y).astype(F16)
            c16 = F16(2.0 * float(self.s2))
            inv16 = F16(F32(1.0) / F32(c16))
            out = np.empty((X16.shape[0], Y16.shape[0]))
            sign = 1.0 if opx != "I" else -1.0
            op = opx if opx != "I" else opy
            for i0 in range(0, X16.shape[0], 128):
                r = X16[i0:i0 + 128, None, :] - Y16[None, :, :]
                sq = r * r
                S = sq.astype(F32).sum(axis=2, dtype=F32).astype(F16)
                if variant == "recip":
                    q = ((-S).astype(F32) * F32(inv16)).astype(F16)
                else:
                    q = ((-S).astype(F32) / F32(c16)).astype(F16)
                kap = np.exp(q.astype(np.float64)).astype(F32).astype(F16)
                if op == "I":
                    out[i0:i0 + 128] = kap.astype(np.float64); continue
                if variant == "recip":
                    t1 = (kap.astype(F32) * F32(inv16)).astype(F16)
                else:
                    t1 = (kap.astype(F32) / F32(c16)).astype(F16)
                if op == "dt":
                    g = ((-t1).astype(F32) * (F16(2.0) * r[:, :, d]).astype(F32)).astype(F16)
                else:
                    gk = ((-t1)[:, :, None].astype(F32) * (F16(2.0) * r[:, :, :d]).astype(F32)).astype(F16)
                    g = gk.astype(F32).sum(axis=2, dtype=F32).astype(F16)
                out[i0:i0 + 128] = sign * g.astype(np.float64)
            return out
    return G

for d in (20, 40, 60, 80):
    head = LOG["quadrature"][str(d)]["simple_uniform"]["head"]
    want = float([l for l in head if l.startswith("GP rel L2")][0].split("->")[1])
    l1 = [l for l in head if l.startswith("GP L1")][0]
    np.random.seed(1234)
    dom, bdy = deepxde_points(d, 1000, 200)
    xt = np.concatenate(deepxde_points(d, 1000, 200))
    ex = logistic_wave_f16(xt).astype(np.float64)[:, 0]
    for variant in ("div", "recip"):
        gp = make(variant)(GradDependentNonlinear(d + 1), reference_laplacian_idx(d, "partitionable"), f16_graph=True)
        gp.GPsolver(dom.astype(np.float64), bdy.astype(np.float64), GN_steps=20)
        err = np.abs(gp.predict(xt.astype(np.float64))[:, 0] - ex)
        rel = np.linalg.norm(err) / np.linalg.norm(ex)
        print(d, variant, 'rel %.10f diff %+.2e  L1 max %.6f mean %.8f' % (rel, rel - want, err.max(), err.mean()), flush=True)
    print('   logged', want, l1[14:80])
