"""Element-wise parity of the as-coded MATRIX-CORE path against the oracle, with every discrepancy accounted for.

Why a plain tolerance is not enough here: with the as-coded surrogate u_hat and eps_PDE leave the evaluation as float16 VALUES
(models/GP.py:671, 769).  An entry whose float16 rounding is decided on a float32 number (matrix-core kernel) instead of a float64 one
(oracle) can land on the neighbouring float16, which moves u_hat / eps_PDE of that tree site by one float16 ulp (2.4e-4 .. 4.9e-4) and a z
component of its root by that times w_k N / (MC delta_t) (solvers/ScaSML.py:248-254, 275-280) -- up to ~1e-2 in the full-history variant.
A blanket bound of that size would also pass an indexing bug confined to small z components.  Instead three device runs on the SAME roots
and random stream account for everything:

  A  the product path: GENERATE -> scasml_gp_eval_compat_sites (matrix cores) -> ACCUMULATE
  B  the same with the float64 evaluation kernel (scasml_gp_eval_compat: roundings decided as NumPy decides them)
  H  ACCUMULATE on B's surrogate values in which ONLY the u_hat / eps_PDE entries that differ between A and B are taken from A

and the assertions are
  1. B equals the oracle to the tolerance of the un-rounded surrogate (5e-5 + 2e-4 |want|) on >= 99 % of the elements and to 2e-4 + 2e-4 |want|
     on all of them (measured: 6e-7 .. 8e-5 -- B's tree points are float32, the oracle's float64, so even B's final float16 rounding of a
     u_hat lands on the other side once in a thousand values);
  2. A and B differ in u_hat / eps_PDE at some of the CONSUMED site values (measured: 14 % at d = 100 with 1200 collocation points, 33 % at
     d = 250 with 2000, more where the value is small and its float16 grid fine: some ten of a value's ~5000 float16-rounded entries round the
     other way on a float32 number, each moving the un-rounded sum by ~2e-5 against a final ulp of 2.4e-4 at 0.5), each by the size of that rounding noise: <= 1e-3 absolute (measured <= 5e-4, typically one float16 ulp of a value near 0.5), as does div u_hat;
  3. A equals H to 2e-6 + 1e-5 |z|, every element (measured: 4e-8 .. 2e-7): the flipped roundings of (2) are the ONLY thing that separates the
     product path from the oracle-exact one -- the random stream, the site order and the accumulation are then identical;
  4. negative controls on copies of the surrogate values: dropping eps_PDE at the sites that consume it, or handing the sites whose u_hat and
     div u_hat enter f the u_hat of their neighbour site, is caught by (3)'s comparison.
What a site consumes (scasml_plan_site_kinds): kind 0 -- Euler-Maruyama sites of level-0 terms -- eps_PDE only (their defect
f(u_hat + 0, ..) - f(u_hat, ..) vanishes, ScaSML.py:43-47 with uz_solve(0) = 0); kind 4 -- sites of higher-level terms -- u_hat and div u_hat;
kinds 1 and 3 -- the root row and terminal samples -- u_hat.
"""
import numpy as np

def _oracle_close(z, want):
    d = np.abs(z - want)
    return bool(np.all(d <= 2e-4 + 2e-4 * np.abs(want))) and float((d <= 5e-5 + 2e-4 * np.abs(want)).mean()) >= 0.99


def _accounted(a, b):
    return bool(np.all(np.abs(a - b) <= 2e-6 + 1e-5 * np.abs(b)))


def assert_explained(eng, n, par, x_rows, root0, stream_id, want, report=None):
    """x_rows: (k, d+1) float32 roots whose global index starts at root0; want: the oracle's (k, 1+d) for them."""
    import torch
    gp = eng.gp
    assert gp.compat == "reference" and gp._compat_model is not None and gp.compat_eval == "mfma"
    k = len(x_rows)
    x_dev = torch.from_numpy(np.ascontiguousarray(x_rows, dtype=np.float32)).cuda()
    method = type(gp)._eval_rows
    seen = {}

    def capture(tag):
        def f(pts, n_rows, rows_per_site, kinds, out4, x_bound=0.0):
            method(gp, pts, n_rows, rows_per_site, kinds, out4, x_bound=x_bound)
            seen[tag] = (out4[:n_rows].clone(), int(rows_per_site), kinds.clone())
        return f

    def inject(vals):
        def f(pts, n_rows, rows_per_site, kinds, out4, x_bound=0.0):
            out4[:n_rows].copy_(vals)
        return f

    def solve():
        return eng.solve(n, par, x_dev, root0=root0, stream_id=stream_id)[0].cpu().numpy().astype(np.float64)

    try:
        gp._eval_rows = capture("A")
        zA = solve()
        gp.compat_eval = "float64"
        gp._eval_rows = capture("B")
        zB = solve()
        gp.compat_eval = "mfma"
        vA, stride, kinds = seen["A"]
        vB = seen["B"][0]
        sites = kinds.numel()
        assert vA.shape == vB.shape == (sites * stride, 4)
        a, b = vA.view(sites, stride, 4)[:, :k], vB.view(sites, stride, 4)[:, :k]
        kd = kinds.view(sites, 1).expand(sites, k)
        uses_u = (kd == 1) | (kd == 3) | (kd == 4)
        uses_eps = kd == 0
        flip_u = uses_u & (a[..., 0] != b[..., 0])
        flip_e = uses_eps & (a[..., 2] != b[..., 2])
        # (2) how many, and how large
        n_used = int(uses_u.sum()) + int(uses_eps.sum())
        n_flip = int(flip_u.sum()) + int(flip_e.sum())
        # u_hat, eps_PDE (float16 values) and div u_hat are sums of ~5000 float16-rounded entries that largely cancel: what separates their two
        # statements is the entries' rounding noise, absolute in size (measured <= 2.5e-4), not an ulp of the (possibly small) result
        worst = float((a[..., 0] - b[..., 0]).abs()[flip_u].max()) if bool(flip_u.any()) else 0.0
        worst_eps = float((a[..., 2] - b[..., 2]).abs()[flip_e].max()) if bool(flip_e.any()) else 0.0
        uses_div = kd == 4
        ddiv = float((a[..., 1] - b[..., 1]).abs()[uses_div].max()) if bool(uses_div.any()) else 0.0
        div_scale = max(1.0, float(b[..., 1].abs()[uses_div].max())) if bool(uses_div.any()) else 1.0
        # (3) B's values with A's u_hat / eps_PDE at the flipped entries only
        h = vB.clone().view(sites, stride, 4)
        h[:, :k, 0] = torch.where(flip_u, a[..., 0], b[..., 0])
        h[:, :k, 2] = torch.where(flip_e, a[..., 2], b[..., 2])
        gp._eval_rows = inject(h.view(-1, 4))
        zH = solve()
        # (4) negative controls
        drop = h.clone()
        drop[:, :k, 2] = torch.where(uses_eps, torch.zeros_like(a[..., 2]), drop[:, :k, 2])
        gp._eval_rows = inject(drop.view(-1, 4))
        z_drop = solve()
        shift = h.clone()
        em = torch.nonzero(kinds == 4).flatten()                  # u_hat of every site that feeds f taken from the next such site
        shift[em, :k, 0] = h[torch.roll(em, -1), :k, 0]
        gp._eval_rows = inject(shift.view(-1, 4))
        z_shift = solve()
    finally:
        gp.compat_eval = "mfma"
        if "_eval_rows" in gp.__dict__:
            del gp._eval_rows
    stats = {"roots": k, "sites": sites, "consumed_values": n_used, "flipped": n_flip, "worst_u_hat_flip": worst, "worst_eps_flip": worst_eps, "max_abs_div_A_vs_B": ddiv,
             "max_abs_B_vs_oracle": float(np.abs(zB - want).max()), "max_abs_A_vs_H": float(np.abs(zA - zH).max()),
             "max_abs_A_vs_oracle": float(np.abs(zA - want).max()), "max_abs_drop": float(np.abs(zA - z_drop).max()),
             "max_abs_shift": float(np.abs(zA - z_shift).max())}
    if report is not None:
        report.update(stats)
    print("explained parity:", stats)
    assert worst <= 1e-3 and worst_eps <= 1e-3, stats             # (2)
    assert ddiv <= 1e-3 * div_scale, stats
    assert _oracle_close(zB, want), stats                         # (1)
    assert _accounted(zA, zH), stats                              # (3)
    assert n_flip == 0 or not np.array_equal(zA, zB)              # the flips are what moves A off B
    assert not _accounted(zA, z_drop), stats                      # (4): the comparison has teeth
    assert int((kinds == 4).sum()) < 2 or not _accounted(zA, z_shift), stats
    return zA
