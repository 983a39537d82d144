"""jax.random as the reference's runs used it, restated without JAX.  TEST INFRASTRUCTURE (oracle/__init__.py).

The reference draws every Brownian increment with ``random.normal(subkey, shape, dtype=jnp.float16)`` (solvers/MLP.py:178, 221) under keys
it derives with ``random.split``.  JAX is not installed in this image, but these are integer and float16 operations that its public source
specifies completely, on top of Threefry-2x32 (scasml_gp_amd/threefry.py: pinned by the Random123 vectors, and its "partitionable" layout
-- ``jax_threefry_partitionable``, the default from jax 0.5 -- by the Hutchinson indices the logged GP errors reproduce):

* ``split(key, n)``: key i = Threefry(key, counter (0, i));
* ``random_bits(key, 16, shape)``: element with row-major index i = low 16 bits of (y0 ^ y1), (y0, y1) = Threefry(key, (0, i));
* ``uniform(key, shape, float16, lo, hi)``: ``bits >> 6 | 0x3C00`` viewed as float16 is in [1, 2); minus 1; times ``hi - lo`` plus ``lo``
  (each operation rounded to float16); clamped below by ``lo``;
* ``normal``: ``lo = nextafter(-1, 0)``, ``hi = 1``, ``sqrt(2) * erf_inv(u)`` with erf_inv evaluated in float32 (XLA's ErfInv32: the two
  degree-8 polynomials of M. Giles, "Approximating the erfinv function", in w = -log1p(-x^2)) and rounded to float16 before the product.

That this is the stream the logged runs consumed is shown by tests/test_reference_replay.py: with it, the replay of ``MLP.u_solve``
(oracle/replay.py) reproduces the MLP numbers of results/**/SimpleUniform.log to all sixteen printed digits at d = 20, 40, 60, 80.
"""
import numpy as np

from scasml_gp_amd import threefry as _tf

F16, F32 = np.float16, np.float32
LAYOUT = "partitionable"

_GILES_CENTRAL = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)
_GILES_TAIL = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)


def prng_key(seed):
    """``random.PRNGKey(seed)`` for 0 <= seed < 2**32: the words (0, seed)."""
    return np.array([0, int(seed) & 0xFFFFFFFF], dtype=np.uint64)


def split(key, num=2):
    return _tf.split(key, num, LAYOUT)


def _horner32(coef, w):
    p = np.full_like(w, F32(coef[0]))
    for c in coef[1:]:
        p = (F32(c) + (p * w).astype(F32)).astype(F32)
    return p


def erf_inv32(x):
    """XLA's single-precision erf_inv."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        # log1p and sqrt through float64 and one rounding: the correctly rounded float32 value on every machine (NumPy's float32 SIMD
        # routines may differ in the last place between CPUs; the logged digits are reproduced either way)
        w = (-(np.log1p((-(x * x)).astype(F32).astype(np.float64)).astype(F32))).astype(F32)
        central = _horner32(_GILES_CENTRAL, (w - F32(2.5)).astype(F32))
        tail = _horner32(_GILES_TAIL, (np.sqrt(np.maximum(w, F32(0)).astype(np.float64)).astype(F32) - F32(3.0)).astype(F32))
    out = (np.where(w < F32(5.0), central, tail) * x).astype(F32)
    return np.where(np.abs(x) == 1, np.copysign(F32(np.inf), x), out)


def bits16(key, index):
    """The 16-bit draws at the row-major positions ``index`` of an array drawn under ``key`` (random access: counter-based)."""
    index = np.asarray(index, dtype=np.uint64)
    y0, y1 = _tf.threefry2x32(key, index >> np.uint64(32), index & np.uint64(0xFFFFFFFF))
    return ((y0 ^ y1) & np.uint64(0xFFFF)).astype(np.uint16)


def normal_f16_at(key, index):
    """``random.normal(key, shape, float16)`` at the row-major positions ``index`` (any integer array; the shape itself does not enter)."""
    lo = np.nextafter(F16(-1.0), F16(0.0))
    one_two = ((bits16(key, index) >> np.uint16(6)) | np.uint16(0x3C00)).view(F16)
    u = (one_two - F16(1.0)).astype(F16)
    u = ((u * F16(F16(1.0) - lo)).astype(F16) + lo).astype(F16)
    u = np.maximum(lo, u)
    return (F16(np.sqrt(2.0)) * erf_inv32(u).astype(F16)).astype(F16)


def normal_f16(key, shape):
    n = int(np.prod(shape))
    return normal_f16_at(key, np.arange(n, dtype=np.uint64)).reshape(shape)


def uniform_f16_at(key, index):
    """``random.uniform(key, shape, float16)`` (minval 0, maxval 1) at the row-major positions ``index``: one of the 1024 values k / 1024."""
    one_two = ((bits16(key, index) >> np.uint16(6)) | np.uint16(0x3C00)).view(F16)
    return np.maximum(F16(0.0), (one_two - F16(1.0)).astype(F16))


def uniform_f16(key, shape):
    n = int(np.prod(shape))
    return uniform_f16_at(key, np.arange(n, dtype=np.uint64)).reshape(shape)
