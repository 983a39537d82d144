"""Oracle PDE-constrained Gaussian process.  TEST INFRASTRUCTURE (oracle/__init__.py).

Float64 NumPy restatement of models/GP.py (``GP`` :8-692 and
``GP_Grad_Dependent_Nonlinear`` :693-769) with the nested autodiff replaced by the
closed forms of SURVEY.md Appendix C, written here in the (a, rho^2, S, r_t) variables of
that table (the HIP kernels use a different, factored arrangement -- the two are checked
against each other and against finite differences of kappa in tests/).

Documented deviations (DESIGN.md "Quirk decisions"): the Laplacian blocks are the exact
spatial Laplacian, not the 5-index subsample on a permuted vector (models/GP.py:28-39,
87-105; E-6, not reproducible: the index set comes from threefry); the factor is a
Cholesky of K + nugget*I, which equals the reference's U*sqrt(S+nugget) product
(models/GP.py:260-267; E-7) for symmetric PSD K; nothing is rounded to float16; the
Newton start is 0 instead of 1e-3 * N(0,1) from PRNGKey(0) (models/GP.py:501).
"""
import numpy as np


class OracleGP:
    def __init__(self, eq):
        self.eq = eq
        self.d = eq.d
        self.sigma_eq = eq.sigma()
        self.s2 = (eq.sigma() * np.sqrt(self.d)) ** 2      # models/GP.py:25  kernel variance sigma^2
        self.a = 1.0 / self.s2
        self.nugget = 1e-2                                  # models/GP.py:26

    # ---------------------------------------------------------------- pair geometry
    def _pairs(self, X, Y):
        X = np.asarray(X, dtype=np.float64)
        Y0 = Y
        Y = np.asarray(Y, dtype=np.float64)
        # the collocation side (row norms, coordinate sums) is kept per array: same numbers, computed once (a solver calls this point by point)
        cache = self.__dict__.setdefault("_pcache", {})
        hit = cache.get(id(Y0))
        if hit is None or hit[0] is not Y0:
            if len(cache) > 8:
                cache.clear()
            hit = cache[id(Y0)] = (Y0, Y, (Y * Y).sum(1), Y[:, :-1].sum(1), np.ascontiguousarray(Y[:, -1]))
        _, Y, y2, ys, yt = hit
        diff2 = (X * X).sum(1)[:, None] + y2[None, :] - 2.0 * X @ Y.T
        rt = X[:, -1][:, None] - yt[None, :]
        S = X[:, :-1].sum(1)[:, None] - ys[None, :]
        rho2 = diff2 - rt * rt
        kap = np.exp(-self.a * diff2 / 2.0)                 # models/GP.py:41-43
        return kap, rho2, S, rt

    def block(self, opx, opy, X, Y):
        """(L^opx_x L^opy_y kappa)(X_i, Y_j); ops in {"I","lap","dt","div"} (Appendix C)."""
        a, d = self.a, self.d
        kap, rho2, S, rt = self._pairs(X, Y)
        lap = a * a * rho2 - a * d
        key = (opx, opy)
        if key == ("I", "I"):
            P = 1.0
        elif key == ("I", "lap") or key == ("lap", "I"):
            P = lap                                         # :87-105 (exact form)
        elif key == ("I", "dt"):
            P = a * rt                                      # :69-73
        elif key == ("dt", "I"):
            P = -a * rt                                     # :59-63
        elif key == ("I", "div"):
            P = a * S                                       # :81-85
        elif key == ("div", "I"):
            P = -a * S                                      # :75-79
        elif key == ("dt", "dt"):
            P = a - a * a * rt * rt                         # :107-111
        elif key == ("dt", "div") or key == ("div", "dt"):
            P = -a * a * rt * S                             # :113-117, 129-133
        elif key == ("div", "div"):
            P = a * d - a * a * S * S                       # :135-139
        elif key == ("dt", "lap"):
            P = -a * rt * lap                               # :119-127
        elif key == ("lap", "dt"):
            P = a * rt * lap                                # :151-159
        elif key == ("div", "lap"):
            P = -(a * S * lap - 2 * a * a * S)              # :141-149
        elif key == ("lap", "div"):
            P = a * S * lap - 2 * a * a * S                 # :161-169
        elif key == ("lap", "lap"):
            P = a ** 4 * rho2 ** 2 - (2 * d + 4) * a ** 3 * rho2 + (d * d + 2 * d) * a * a   # :171-179
        else:
            raise KeyError(key)
        return P * kap

    # ---------------------------------------------------------------- Gram (models/GP.py:182-268)
    _ROWS = (("I", "dom"), ("I", "bdy"), ("lap", "dom"), ("dt", "dom"), ("div", "dom"))

    def kernel_phi_phi(self, x_dom, x_bdy):
        self.x_t_domain = np.asarray(x_dom, dtype=np.float64)
        self.x_t_boundary = np.asarray(x_bdy, dtype=np.float64)
        self.N_domain, self.N_boundary = len(self.x_t_domain), len(self.x_t_boundary)
        self.phi_dim = 4 * self.N_domain + self.N_boundary
        pts = {"dom": self.x_t_domain, "bdy": self.x_t_boundary}
        K = np.block([[self.block(ox, oy, pts[px], pts[py]) for (oy, py) in self._ROWS]
                      for (ox, px) in self._ROWS])          # :251-258 block order
        return K

    # ---------------------------------------------------------------- operator F (models/GP.py:705-743)
    def time_der_rep(self, sol):
        """F(z) of models/GP.py:705-719: for Grad_Dependent_Nonlinear -s^2 z1 z5 + (1/d + s^2/2) z5 - (s^2/2) z3; in general
        u_t = -mu div u - sigma^2/2 Lap u - f(u, sigma div u) (oracle/equation.py, F_parts)."""
        N = self.N_domain
        return self.eq.F_parts(sol[:N], sol[N:2 * N], sol[2 * N:])[0]

    def _b(self, sol, bdy_g):
        N = self.N_domain
        return np.concatenate([sol[:N], bdy_g, sol[N:2 * N], self.time_der_rep(sol), sol[2 * N:]])

    # ---------------------------------------------------------------- training (models/GP.py:487-604)
    def GPsolver(self, x_dom, x_bdy, GN_steps=20):
        K = self.kernel_phi_phi(x_dom, x_bdy)
        M = self.phi_dim
        Kp = K + self.nugget * np.eye(M)                    # :260-267 (== L L^T of the SVD factor)
        self.cholesky_phi_phi_perturb = np.linalg.cholesky(Kp)
        A = np.linalg.inv(Kp)
        A = 0.5 * (A + A.T)
        self._newton(A, GN_steps)
        z = self._b(self.sol, self._bdy_g)                  # :593-598
        self.right_vector = np.linalg.solve(Kp, z)[:, None]  # :599-600
        return self.predict(self.x_t_domain)                # :602

    def _g_boundary(self):
        return self.eq.g(self.x_t_boundary)[:, 0]

    def _newton(self, A, GN_steps):
        """Newton iteration of models/GP.py:501-588 on J(sol) = b(sol)^T A b(sol); A = (L L^T)^-1."""
        M, N, Nb = self.phi_dim, self.N_domain, self.N_boundary
        bdy_g = self._bdy_g = self._g_boundary()                           # :417-419
        d, s = self.d, self.sigma_eq
        r1, r3, r4, r5 = slice(0, N), slice(N + Nb, 2 * N + Nb), slice(2 * N + Nb, 3 * N + Nb), slice(3 * N + Nb, M)
        sol = np.zeros(3 * N)
        damping = 1e-4                                      # :490
        hist = []

        def loss(sol_):
            b = self._b(sol_, bdy_g)
            return float(b @ A @ b)                         # :430-444  (|L^-1 b|^2 = b^T Kp^-1 b)

        hist.append(loss(sol))
        for _ in range(GN_steps):                           # :515-588
            b = self._b(sol, bdy_g)
            Ab = A @ b
            # Jacobian of b w.r.t. (z1, z3, z5): identity blocks + diagonal blocks from F (:722-743)
            _, (dF1, dF3, dF5), (F11, F15, F55) = self.eq.F_parts(sol[:N], sol[N:2 * N], sol[2 * N:])
            grad = 2.0 * np.concatenate([Ab[r1] + dF1 * Ab[r4], Ab[r3] + dF3 * Ab[r4], Ab[r5] + dF5 * Ab[r4]])
            if np.linalg.norm(grad) < 1e-5:                 # :521
                break
            rows = (r1, r3, r5)
            dF = (dF1, dF3, dF5)
            H = np.empty((3 * N, 3 * N))
            for i in range(3):
                for j in range(3):
                    blk = (A[rows[i], rows[j]] + dF[i][:, None] * A[r4, rows[j]]
                           + A[rows[i], r4] * dF[j][None, :] + dF[i][:, None] * A[r4, r4] * dF[j][None, :])
                    H[i * N:(i + 1) * N, j * N:(j + 1) * N] = 2.0 * blk
            # second-order terms: the Hessian of F_i in (z1_i, z5_i) (for Grad_Dependent_Nonlinear only the mixed one, -sigma^2),
            # weighted by 2*(A b)_{F_i}
            idx = np.arange(N)
            H[idx, idx] += 2.0 * F11 * Ab[r4]
            H[idx, 2 * N + idx] += 2.0 * F15 * Ab[r4]
            H[2 * N + idx, idx] += 2.0 * F15 * Ab[r4]
            H[2 * N + idx, 2 * N + idx] += 2.0 * F55 * Ab[r4]
            step = np.linalg.solve(H + damping * np.eye(3 * N), -grad)   # :529-533
            sol = sol + step                                # alpha = 1, :541,573
            hist.append(loss(sol))
        self.loss_history = hist
        self.sol = sol

    # ---------------------------------------------------------------- inference (models/GP.py:630-687, 746-769)
    def _features(self, opx, X):
        dom, bdy = self.x_t_domain, self.x_t_boundary
        return np.concatenate([self.block(opx, "I", X, dom), self.block(opx, "I", X, bdy),
                               self.block(opx, "lap", X, dom), self.block(opx, "dt", X, dom),
                               self.block(opx, "div", X, dom)], axis=1)       # :643-649 order

    def predict(self, X):
        return self._features("I", X) @ self.right_vector

    def pde_parts(self, X):
        rv = self.right_vector
        return (self._features("dt", X) @ rv, self._features("div", X) @ rv, self._features("lap", X) @ rv)

    def compute_PDE_loss(self, X):
        """dt u + mu div u + sigma^2/2 Lap u + f(u, sigma div u); for Grad_Dependent_Nonlinear this is models/GP.py:767-768,
        dt + (s^2 u - 1/d - s^2/2) div + s^2/2 lap."""
        s = self.sigma_eq
        dt, div, lap = self.pde_parts(X)
        sol = self.predict(X)
        return dt + self.eq.mu() * div + (s ** 2 / 2) * lap + self.eq.f_parts(sol, s * div)[0]

    def compute_gradient(self, X, sol=None):
        """Full gradient (N, d+1) of the posterior mean, time derivative last (:673-687).
        d/dx_i [P(rho^2,S,r_t) kappa] = (dP/d rho^2 * 2 r_i + dP/dS - a r_i P) kappa."""
        X = np.asarray(X, dtype=np.float64)
        a, d = self.a, self.d
        dom, bdy = self.x_t_domain, self.x_t_boundary
        N, Nb = self.N_domain, self.N_boundary
        rv = self.right_vector[:, 0]
        c0, c2 = rv[:N], rv[N:N + Nb]
        cL, ct, cS = rv[N + Nb:2 * N + Nb], rv[2 * N + Nb:3 * N + Nb], rv[3 * N + Nb:]
        out = np.zeros((X.shape[0], d + 1))
        # boundary features: plain kappa
        kap, _, _, rt = self._pairs(X, bdy)
        wk = kap * c2[None, :]
        out[:, :d] += -a * (X[:, :d] * wk.sum(1)[:, None] - wk @ bdy[:, :d])
        out[:, d] += (-a * rt * wk).sum(1)
        # domain features: E = c0 + cL*lap + ct*a*rt + cS*a*S
        kap, rho2, S, rt = self._pairs(X, dom)
        lap = a * a * rho2 - a * d
        E = c0[None, :] + cL[None, :] * lap + ct[None, :] * a * rt + cS[None, :] * a * S
        coef_r = kap * (2 * a * a * cL[None, :] - a * E)     # multiplies r_i = x_i - y_i
        coef_1 = kap * (a * cS[None, :])                     # dS/dx_i = 1
        out[:, :d] += X[:, :d] * coef_r.sum(1)[:, None] - coef_r @ dom[:, :d] + coef_1.sum(1)[:, None]
        out[:, d] += (kap * (a * ct[None, :] - a * rt * E)).sum(1)
        return out
