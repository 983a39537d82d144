"""Oracle RNG: Philox4x32-10 + a bit-reproducible Box-Muller normal generator.

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
* Normals: Box-Muller on 24-bit uniforms, with ln / sin / cos evaluated by fixed
  polynomials (Cephes single-precision coefficients) in Horner form, every step ONE
  IEEE-754 binary32 operation: a multiply, an add, a correctly-rounded sqrt, or -- where
  the code below says ``fma32`` -- a fused multiply-add (one rounding; ``fmaf`` on the GPU,
  an exact emulation through float64 here).  Any conforming implementation therefore
  produces the same bits.  (Round 1 specified separately rounded multiply and add
  throughout; the fused form is 16 % fewer instructions in a kernel that is bound by them.)
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
_LOG_P = [_f32(v) for v in (7.0376836292e-2, -1.1514610310e-1, 1.1676998740e-1, -1.2420140846e-1,
                            1.4249322787e-1, -1.6668057665e-1, 2.0000714765e-1, -2.4999993993e-1,
                            3.3333331174e-1)]
_LN2_HI = _f32(0.693359375)
_LN2_LO = _f32(-2.12194440e-4)
_SQRT2 = _f32(1.41421354)
_SIN_P = [_f32(v) for v in (-1.9515295891e-4, 8.3321608736e-3, -1.6666654611e-1)]
_COS_P = [_f32(v) for v in (2.443315711809948e-5, -1.388731625493765e-3, 4.166664568298827e-2)]
_ANGLE_SCALE = _f32(np.pi / 2 * 2.0 ** -22)  # (pi/2) * 2^-22 rounded to binary32
_HALF = _f32(0.5)
_ONE = _f32(1.0)
_TWO_M24 = _f32(2.0 ** -24)


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


def ln_u24(k):
    """ln(k * 2^-24) for integer k in [1, 2^24], binary32, mul/add only (Cephes logf scheme)."""
    f = k.astype(np.float32)                       # exact: k <= 2^24
    bits = f.view(np.uint32)
    e = (bits >> np.uint32(23)).astype(np.int32) - np.int32(127)
    m = ((bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)  # [1, 2)
    big = m > _SQRT2
    m = np.where(big, m * _HALF, m)                # exact
    e = np.where(big, e + np.int32(1), e)
    x = m - _ONE                                   # exact (Sterbenz)
    z = x * x
    p = np.broadcast_to(_LOG_P[0], x.shape)
    for c in _LOG_P[1:]:
        p = fma32(p, x, c)
    y = x * z
    y = y * p
    fe = (e - np.int32(24)).astype(np.float32)
    y = fma32(fe, _LN2_LO, y)
    y = fma32(-_HALF, z, y)
    r = x + y
    r = fma32(fe, _LN2_HI, r)
    return r.astype(np.float32)


def sincos_u24(k):
    """(cos, sin) of a uniformly distributed angle built from the 24-bit integer k.

    The top two bits choose the quadrant; the low 22 bits give an angle in
    (-pi/4, pi/4) evaluated by the Cephes sinf/cosf kernels.  (The angle carries a
    constant pi/4 phase relative to 2*pi*k/2^24, irrelevant for a uniform angle.)
    """
    quad = (k >> np.uint32(22)).astype(np.int32)
    frac = (k & np.uint32(0x3FFFFF)).astype(np.int32)
    w = (frac - np.int32(1 << 21)).astype(np.float32) + _HALF    # exact half-integers
    x = w * _ANGLE_SCALE
    z = x * x
    s = fma32(_SIN_P[0], z, _SIN_P[1])
    s = fma32(s, z, _SIN_P[2])
    s = s * z
    s = fma32(s, x, x)
    c = fma32(_COS_P[0], z, _COS_P[1])
    c = fma32(c, z, _COS_P[2])
    c = fma32(c, z * z, fma32(-_HALF, z, _ONE))
    cc = np.where(quad == 0, c, np.where(quad == 1, -s, np.where(quad == 2, -c, s)))
    ss = np.where(quad == 0, s, np.where(quad == 1, c, np.where(quad == 2, -s, -c)))
    return cc.astype(np.float32), ss.astype(np.float32)


def box_muller(ra, rb):
    """Two N(0,1) binary32 values from two uint32 words."""
    k1 = (ra >> np.uint32(8)) + np.uint32(1)
    k2 = rb >> np.uint32(8)
    t = _f32(-2.0) * ln_u24(k1)
    t = np.where(t < 0, _f32(0.0), t)
    rad = np.sqrt(t).astype(np.float32)
    c, s = sincos_u24(k2)
    return (rad * c).astype(np.float32), (rad * s).astype(np.float32)


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
    n0, n1 = box_muller(r0, r1)
    n2, n3 = box_muller(r2, r3)
    out = np.stack([n0, n1, n2, n3], axis=-1).reshape(root.shape[0], 4 * nq)
    return out[:, :d]


def uniform_tau(seed, stream, root, site):
    """One U(0,1) binary32 value per root for the full-history time draw: ((r0>>9)+0.5)*2^-23."""
    root = np.asarray(root, dtype=np.uint64)
    r0, _, _, _ = philox4x32_10(np.uint64(QUAD_TAU), np.uint64(site), root, np.uint64(stream),
                                np.uint64(seed) & _MASK, np.uint64(seed) >> _S32)
    v = (r0 >> np.uint32(9)).astype(np.float32) + _HALF
    return (v * _f32(2.0 ** -23)).astype(np.float32)
