"""Oracle RNG: Philox4x32-10 + a bit-reproducible inverse-CDF normal generator.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Replaces the reference's
``jax.random.normal(key, (B, MC, d), float16)`` / ``random.uniform`` draws
(solvers/MLP.py:167-168,178,220-221; solvers/MLP_full_history.py:92-93,99,133,138).
JAX's threefry stream is an un-vendored dependency and cannot be reproduced here
(SURVEY.md Appendix E-2/E-3), so the build defines its own counter-based stream:

* Philox4x32-10 exactly as published (Salmon et al., SC'11; Random123 v1.14
  ``philox.h``): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments
  0x9E3779B9 / 0xBB67AE85, ten rounds.  Checked against the Random123 known-answer
  vectors in tests/test_oracle_philox.py.
* Counter layout  (c0, c1, c2, c3) = (quad, site, root, stream),  key = 64-bit seed.
  ``quad`` q yields the four normals of spatial dims 4q..4q+3; ``site`` is the static
  index of the path-step inside one root's Picard tree (oracle/mlp.py ``site_count``);
  ``root`` is the global index of the evaluation point; ``stream`` separates solver
  calls.  quad = 0x80000000 is reserved for the full-history uniform time draw.
* Normals: one per Philox word, the inverse normal CDF of the 24-bit uniform in the word's top bits, evaluated as a
  per-segment cubic from a 768-row table (``normal_table``) with three fused multiply-adds in IEEE-754 binary32
  (``fma32``: one rounding; ``fmaf`` on the GPU, an exact emulation through float64 here).  Integer bit manipulation,
  a table row and three correctly rounded operations: any conforming implementation produces the same bits.
  (Round 1 and most of round 2 used Box-Muller with Cephes polynomials; the table form is half the vector instructions
  in kernels that are bound by them.)
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint64(0x9E3779B9)
_W1 = np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)

QUAD_TAU = 0x80000000  # c0 value reserved for the full-history uniform draw


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. Inputs broadcastable unsigned ints; returns 4 uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
        *[np.asarray(v).astype(np.uint64) & _MASK for v in (c0, c1, c2, c3, k0, k1)])
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


_f32 = np.float32
_HALF = _f32(0.5)


def fma32(a, b, c):
    """Correctly rounded binary32 fused multiply-add a*b + c, vectorised.  The product of two binary32 numbers is exact
    in binary64; the binary64 sum is rounded once more, which can differ from the single binary32 rounding only when it
    lands EXACTLY on a binary32 midpoint while the true sum does not -- detected from the sum's exact error (TwoSum) and
    resolved by stepping off the midpoint in the error's direction before the final rounding."""
    a64, b64, c64 = (np.asarray(v, dtype=np.float32).astype(np.float64) for v in (a, b, c))
    a64, b64, c64 = np.broadcast_arrays(a64, b64, c64)
    prod = a64 * b64
    s = prod + c64
    bb = s - prod
    err = (prod - (s - bb)) + (c64 - bb)
    mid = (np.ascontiguousarray(s).view(np.uint64) & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
    fix = mid & (err != 0)
    s = np.where(fix, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


_TABLE = None


def normal_table():
    """The 768 x 4 binary32 coefficients of the inverse normal CDF, restated from the definition (the product carries them
    as a committed constant, scasml_gp_amd/csrc/normal_table.inc; tests/test_oracle_philox.py and tests/test_gpu_rng.py check
    that the two agree bit for bit).  Row 32 e + g covers the odd integers v = 2 j + 1 in [2^e (1 + g/32), 2^e (1 + (g+1)/32)),
    i.e. p = v 2^-25; its cubic is the Hermite interpolant of Phi^-1 between the segment's ends (values from scipy's ndtri,
    slopes 1 / phi), in the integer coordinate s = low 18 mantissa bits of binary32(v)."""
    global _TABLE
    if _TABLE is None:
        from scipy.special import ndtri
        e = np.repeat(np.arange(24, dtype=np.float64), 32)
        g = np.tile(np.arange(32, dtype=np.float64), 24)
        pa = 2.0 ** e * (1.0 + g / 32.0) * 2.0 ** -25
        pb = 2.0 ** e * (1.0 + (g + 1.0) / 32.0) * 2.0 ** -25
        xa, xb = ndtri(pa), ndtri(pb)
        ha = (pb - pa) * np.sqrt(2.0 * np.pi) * np.exp(0.5 * xa * xa)
        hb = (pb - pa) * np.sqrt(2.0 * np.pi) * np.exp(0.5 * xb * xb)
        dx = xb - xa
        _TABLE = np.stack([xa, ha * 2.0 ** -18, (3.0 * dx - 2.0 * ha - hb) * 2.0 ** -36,
                           (ha + hb - 2.0 * dx) * 2.0 ** -54], axis=1).astype(np.float32)
    return _TABLE


def icdf_normal(r):
    """One N(0,1) binary32 value per uint32 word: inverse CDF of u = ((r >> 8) + 1/2) 2^-24 by table (see normal_table)."""
    r = np.asarray(r, dtype=np.uint32)
    upper = (r >> np.uint32(31)).astype(bool)                       # u > 1/2
    k = r >> np.uint32(8)
    j = np.where(upper, k ^ np.uint32(0xFFFFFF), k) & np.uint32(0x7FFFFF)
    b = (np.uint32(2) * j + np.uint32(1)).astype(np.float32).view(np.uint32)   # exact: < 2^24
    c = normal_table()[(b >> np.uint32(18)) - np.uint32(127 << 5)]
    s = (b & np.uint32(0x3FFFF)).astype(np.float32)                 # exact
    x = fma32(fma32(fma32(c[..., 3], s, c[..., 2]), s, c[..., 1]), s, c[..., 0])
    return np.where(upper, -x, x).astype(np.float32)


def normals(seed, stream, root, site, d):
    """Standard normals for one path-step ``site`` of every root in ``root``.

    root: (B,) uint array of global root indices.  Returns float32 (B, d).
    Dim 4q+j of the step is output j of the Philox block with c0 = q.
    """
    root = np.asarray(root, dtype=np.uint64)
    nq = (d + 3) // 4
    q = np.arange(nq, dtype=np.uint64)[None, :]
    r0, r1, r2, r3 = philox4x32_10(q, np.uint64(site), root[:, None], np.uint64(stream),
                                   np.uint64(seed) & _MASK, np.uint64(seed) >> _S32)
    out = np.stack([icdf_normal(r) for r in (r0, r1, r2, r3)], axis=-1).reshape(root.shape[0], 4 * nq)
    return out[:, :d]


def uniform_tau(seed, stream, root, site):
    """One U(0,1) binary32 value per root for the full-history time draw: ((r0>>9)+0.5)*2^-23."""
    root = np.asarray(root, dtype=np.uint64)
    r0, _, _, _ = philox4x32_10(np.uint64(QUAD_TAU), np.uint64(site), root, np.uint64(stream),
                                np.uint64(seed) & _MASK, np.uint64(seed) >> _S32)
    v = (r0 >> np.uint32(9)).astype(np.float32) + _HALF
    return (v * _f32(2.0 ** -23)).astype(np.float32)
