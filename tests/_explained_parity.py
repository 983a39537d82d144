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
  2. A and B differ in u_hat / eps_PDE at some of the CONSUMED site values, and HOW MANY is predicted, not merely observed: two further runs A', B'
     return the same sums before their final float16 rounding (round16 bit 1 off).  Then (a) A = float16(A') and B = float16(B') bit for bit -- the
     un-rounded runs are the same sums (B: but for the float64 kernel's direct rounding, <= 1e-3 of the values); (b) delta = A' - B' is the rounding noise of the ENTRIES (some ten of a value's ~5000 float16-rounded
     entries round the other way on a float32 number, each moving the sum by ~2e-5): sized against the entries' own rounding perturbation A' - A''
     (A'': the same sums, entries not rounded; model ratio sqrt(12 p) = 0.15 .. 0.22, asserted <= 0.35) and UNBIASED (|mean| <= 4 standard errors),
     which a u_hat whose sum is off by a float16 ulp at every site breaks; (c) a value flips where a float16 rounding boundary falls between A' and B', which for a
     boundary placed at random has probability min(1, |delta| / ulp16(value)): the measured share of flipped values must equal the mean of that
     prediction to 25 % + 4 standard errors, PER SITE KIND (u_hat at terminal / root sites, u_hat at level l > 0 sites, eps_PDE at level-0 sites;
     for eps_PDE delta includes sigma^2 div (float16(u_hat) - u_hat), the change of f through the rounded u_hat it is formed from, models/GP.py:767-769);
     (d) each flip moves the value by at most |delta| + one float16 ulp, and div u_hat (never rounded) by the noise bound;
  3. A equals H to 2e-6 + 1e-5 |z|, every element (measured: 4e-8 .. 2e-7): the flipped roundings of (2) are the ONLY thing that separates the
     product path from the oracle-exact one -- the random stream, the site order and the accumulation are then identical;
  4. negative controls on copies of the surrogate values: dropping eps_PDE at the sites that consume it, or handing the sites whose u_hat and
     div u_hat enter f the u_hat of their neighbour site, is caught by (3)'s comparison; adding ONE float16 ulp at every level l > 0 site is caught -- added to the
     rounded u_hat by (2a) (and by (2c)'s flip-rate bound where the noise is well below an ulp), added to the sums themselves by (2b)'s bias bound.
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


def _ulp16(v):
    """float16 ulp at |v| (torch tensor): 2^(e - 10) for a normal value, 2^-24 below 2^-14."""
    import torch
    e = torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -14)))
    return torch.pow(torch.tensor(2.0, dtype=v.dtype, device=v.device), e - 10.0)


# The entries' rounding noise delta = A' - B' of one un-rounded value against the entries' own rounding perturbation: a float16-rounded entry is off
# its exact value by ulp / sqrt(12) rms, and it rounds the OTHER way on a float32 number when that number lies within ~2^-20 relative of a midpoint,
# i.e. with probability p ~ 2 x 2^-20 / 2^-11 = 2^-8 .. 2^-9, moving the sum by a whole ulp: rms(delta) / rms(A' - A'') = sqrt(12 p) = 0.15 .. 0.22,
# A'' being the same matrix-core sums with the entries NOT rounded (round16 = 0); an upper estimate, the float32 value of an entry is usually closer than
# 2^-20.  Measured on MI355X (profiles/r06_explained_parity.txt): 0.06 .. 0.14 from 1200 to 20 000 collocation points, while rms(delta) itself grows 2.4e-5 -> 1.7e-4.  Asserted: ratio <= 0.35, max |delta| <= 8 rms(delta) bound,
# and delta UNBIASED (|mean| <= 4 standard errors: a sum that is off by a float16 ulp at every site has a mean of one ulp).
NOISE_RATIO, NOISE_PEAK = 0.35, 8.0


def _flip_rate_ok(flipped, predicted):
    """measured share of flipped values against the mean predicted flip probability, two-sided: 25 % + 4 standard errors of a share."""
    n = flipped.numel()
    if n == 0:
        return True, 0.0, 0.0
    share, pred = float(flipped.double().mean()), float(predicted.double().mean())
    slack = 0.25 * pred + 4.0 * (max(pred * (1.0 - pred), 1.0 / n) / n) ** 0.5
    return abs(share - pred) <= slack, share, pred


def assert_explained(eng, n, par, x_rows, root0, stream_id, want, report=None):
    """x_rows: (k, d+1) float32 roots whose global index starts at root0; want: the oracle's (k, 1+d) for them."""
    import torch
    gp = eng.gp
    assert gp.compat == "reference" and gp._compat_model is not None and gp.compat_eval == "mfma"
    k = len(x_rows)
    x_dev = torch.from_numpy(np.ascontiguousarray(x_rows, dtype=np.float32)).cuda()
    method = type(gp)._eval_rows
    r16_keep = int(gp.eval_round16)
    seen = {}

    def capture(tag):
        def f(pts, n_rows, rows_per_site, kinds, out4, x_bound=0.0, order=None):
            method(gp, pts, n_rows, rows_per_site, kinds, out4, x_bound=x_bound, order=order)
            seen[tag] = (out4[:n_rows].clone(), int(rows_per_site), kinds.clone())
        return f

    def inject(vals):
        def f(pts, n_rows, rows_per_site, kinds, out4, x_bound=0.0, order=None):
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
        # the same two runs with u_hat and eps_PDE NOT rounded on the way out (round16 bit 1 off; the entries stay rounded)
        r16 = int(gp.eval_round16)
        gp.eval_round16 = r16 & ~2
        gp._eval_rows = capture("A'")
        solve()
        gp.compat_eval = "float64"
        gp._eval_rows = capture("B'")
        solve()
        gp.compat_eval, gp.eval_round16 = "mfma", 0
        gp._eval_rows = capture("A''")                               # the matrix-core sums with NO rounding at all: what the entries' rounding perturbs
        solve()
        gp.eval_round16 = r16
        vA, stride, kinds = seen["A"]
        vB = seen["B"][0]
        sites = kinds.numel()
        assert vA.shape == vB.shape == (sites * stride, 4)
        a, b = vA.view(sites, stride, 4)[:, :k], vB.view(sites, stride, 4)[:, :k]
        au, bu = seen["A'"][0].view(sites, stride, 4)[:, :k], seen["B'"][0].view(sites, stride, 4)[:, :k]
        ax = seen["A''"][0].view(sites, stride, 4)[:, :k]
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
        # (2a) the un-rounded runs are the same sums: their float16 rounding IS the product's value
        # (the float64 kernel rounds its float64 sum to float16 directly, while B' leaves as float32: a double rounding that differs from the
        # direct one once in ~10^4 values -- tolerated at 1e-3 of the values, one float16 ulp each)
        same_a = bool(torch.equal(au[..., 0].half().float()[uses_u], a[..., 0][uses_u]))
        b_off = (bu[..., 0].half().float() != b[..., 0]) & uses_u
        same_b = float(b_off.double().sum()) <= 1e-3 * float(uses_u.sum()) and \
            bool(((bu[..., 0].half().float() - b[..., 0]).abs() <= _ulp16(b[..., 0]) * 1.001)[uses_u].all())
        same_sums = same_a and same_b
        # (2b) the entries' rounding noise
        sig = float(eng.problem().sigma)
        d_u = (au[..., 0] - bu[..., 0]).double()
        d_e = (au[..., 2] - bu[..., 2]).double()
        if int(eng.equation.eq_id) == 0:      # eps_PDE is formed from the ROUNDED u_hat (models/GP.py:767-769): f = sigma u sigma div
            d_e = d_e + sig * sig * (au[..., 1].double() * (a[..., 0] - au[..., 0]).double() - bu[..., 1].double() * (b[..., 0] - bu[..., 0]).double())
        d_div = (a[..., 1] - b[..., 1]).double()
        noise = {"u_hat": d_u[uses_u], "eps_PDE": d_e[uses_eps], "div": d_div[uses_div]}
        # what the entries' rounding does to the same sums (A' - A''), per output; div u_hat is never rounded on the way out: a[..., 1] is A's
        pert = {"u_hat": (au[..., 0] - ax[..., 0]).double()[uses_u], "eps_PDE": (au[..., 2] - ax[..., 2]).double()[uses_eps],
                "div": (a[..., 1] - ax[..., 1]).double()[uses_div]}
        rms = lambda v: float(v.pow(2).mean().sqrt()) if v.numel() else 0.0
        noise_rms = {kk: rms(v) for kk, v in noise.items()}
        noise_max = {kk: float(v.abs().max()) if v.numel() else 0.0 for kk, v in noise.items()}
        pert_rms = {kk: rms(v) for kk, v in pert.items()}
        noise_ratio = {kk: (noise_rms[kk] / pert_rms[kk] if pert_rms[kk] > 0 else 0.0) for kk in noise}
        # bias in standard errors of the mean, after an allowance of 2 % of the perturbation for the float32 summation error of A (div u_hat at d = 100:
        # a mean of 2.4e-6 against a noise of 1.2e-5 rms over 816 values is 5.7 standard errors and 1 % of a float16 ulp)
        noise_bias = {kk: (max(abs(float(v.mean())) - 0.02 * pert_rms[kk], 0.0) / (noise_rms[kk] / v.numel() ** 0.5) if v.numel() > 1 and noise_rms[kk] > 0 else 0.0)
                      for kk, v in noise.items()}
        noise_ok = all(noise_ratio[kk] <= NOISE_RATIO and noise_max[kk] <= NOISE_PEAK * NOISE_RATIO * pert_rms[kk] + 1e-7 and noise_bias[kk] <= 4.0 for kk in noise)
        # (2c) flip rate per site kind against its prediction min(1, |delta| / ulp16)
        classes = {"u_hat at terminal/root sites": ((kd == 1) | (kd == 3), 0, d_u), "u_hat at level l>0 sites": (kd == 4, 0, d_u), "eps_PDE at level-0 sites": (kd == 0, 2, d_e)}
        rates, rate_ok = {}, True
        for name, (mask, col, dl) in classes.items():
            pred = torch.clamp(dl.abs()[mask] / _ulp16(bu[..., col][mask]).double(), max=1.0)
            ok, share, p = _flip_rate_ok((a[..., col] != b[..., col])[mask], pred)
            rates[name] = {"values": int(mask.sum()), "flipped_share": round(share, 4), "predicted_share": round(p, 4), "ok": ok}
            rate_ok = rate_ok and ok
        # (2d) a flip is at most the noise plus one float16 ulp
        lim_u = d_u.abs() + _ulp16(torch.maximum(a[..., 0].abs(), b[..., 0].abs())).double()
        lim_e = d_e.abs() + _ulp16(torch.maximum(a[..., 2].abs(), b[..., 2].abs())).double()
        flip_sized = bool(((a[..., 0] - b[..., 0]).double().abs() <= lim_u * (1 + 1e-6))[uses_u].all()) and \
            bool(((a[..., 2] - b[..., 2]).double().abs() <= lim_e * (1 + 1e-3) + 1e-6)[uses_eps].all())
        # third negative control, in its two forms: ONE float16 ulp added at EVERY level l > 0 site -- a systematic error of the size of a single flip --
        # (i) to the product's rounded u_hat: breaks (2a), and the flip-rate bound wherever the real noise is well below an ulp;
        # (ii) to the un-rounded sums themselves (the rounded values following them): delta gains a mean of one ulp -- breaks (2b)'s bias bound
        m4 = kd == 4
        ctrl_caught, ctrl_detail = True, {}
        if bool(m4.any()):
            one = _ulp16(a[..., 0])
            a_bad = a[..., 0] + torch.where(m4, one, torch.zeros_like(one))
            pred4 = torch.clamp(d_u.abs()[m4] / _ulp16(bu[..., 0][m4]).double(), max=1.0)
            by_rate = not _flip_rate_ok((a_bad != b[..., 0])[m4], pred4)[0]
            by_2a = not bool(torch.equal(au[..., 0].half().float()[m4], a_bad[m4]))
            d_bad = (d_u + torch.where(m4, one, torch.zeros_like(one)).double())[m4]
            bias_bad = max(abs(float(d_bad.mean())) - 0.02 * pert_rms["u_hat"], 0.0) / (rms(d_bad) / d_bad.numel() ** 0.5)
            by_bias = bias_bad > 4.0
            ctrl_detail = {"rounded_plus_one_ulp": {"breaks_2a": by_2a, "breaks_flip_rate": by_rate}, "sums_plus_one_ulp": {"bias_in_standard_errors": round(bias_bad, 1), "breaks_2b": by_bias}}
            ctrl_caught = (by_2a or by_rate) and by_bias
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
        gp.eval_round16 = r16_keep
        if "_eval_rows" in gp.__dict__:
            del gp._eval_rows
    stats = {"roots": k, "sites": sites, "consumed_values": n_used, "flipped": n_flip, "worst_u_hat_flip": worst, "worst_eps_flip": worst_eps, "max_abs_div_A_vs_B": ddiv,
             "max_abs_B_vs_oracle": float(np.abs(zB - want).max()), "max_abs_A_vs_H": float(np.abs(zA - zH).max()),
             "max_abs_A_vs_oracle": float(np.abs(zA - want).max()), "max_abs_drop": float(np.abs(zA - z_drop).max()),
             "max_abs_shift": float(np.abs(zA - z_shift).max()),
             "unrounded_runs_are_the_same_sums": same_sums, "entry_noise_rms": noise_rms, "entry_noise_max": noise_max,
             "entry_rounding_perturbation_rms": pert_rms, "noise_over_perturbation": {kk: round(v, 3) for kk, v in noise_ratio.items()},
             "noise_bias_in_standard_errors": {kk: round(v, 2) for kk, v in noise_bias.items()}, "flip_rates": rates,
             "flips_sized_by_noise_plus_one_ulp": flip_sized, "one_ulp_everywhere_control_caught": ctrl_caught, "controls": ctrl_detail}
    if report is not None:
        report.update(stats)
    print("explained parity:", stats)
    assert same_sums, stats                                       # (2a)
    assert noise_ok, stats                                        # (2b)
    assert rate_ok, stats                                         # (2c)
    assert flip_sized and worst <= 2e-3 and worst_eps <= 2e-3 and max(np.abs(zA - zB).max(), 0.0) < 2e-2, stats             # (2d) + a coarse global cap as a backstop
    assert ctrl_caught, stats                                     # the third control
    assert _oracle_close(zB, want), stats                         # (1)
    assert _accounted(zA, zH), stats                              # (3)
    assert n_flip == 0 or not np.array_equal(zA, zB)              # the flips are what moves A off B
    assert not _accounted(zA, z_drop), stats                      # (4): the comparison has teeth
    assert int((kinds == 4).sum()) < 2 or not _accounted(zA, z_shift), stats
    return zA
