"""The reference's five Hutchinson indices, from the generator it draws them with.

models/GP.py:35 takes ``random.choice(random.PRNGKey(0), self.d, shape=(5,), replace=False)`` inside every
"Laplacian" -- a fixed function of d.  JAX is not installed here, but the draw is integer arithmetic that its public
source and the Random123 paper specify completely:

* ``PRNGKey(0)`` is the key (0, 0);
* ``choice(key, d, (5,), replace=False)`` = ``permutation(key, d)[:5]``, and for d < 2**21 ``permutation`` is ONE round of
  "split the key, draw d 32-bit sort keys from the sub-key, stable-sort arange(d) by them" (jax/_src/random.py ``_shuffle``);
* ``split`` and the 32-bit draws are Threefry-2x32 (20 rounds) of a counter under the key, in one of two counter layouts:
  ``"original"`` (jax < 0.5: counters ``iota(n)`` cut into two halves that form the two input words) and
  ``"partitionable"`` (``jax_threefry_partitionable=True``, the default from jax 0.5: counter i is the 64-bit pair
  (0, i) and the draw is the XOR of the two output words).

The reference pins no JAX version (requirements.txt:8), so both layouts are offered; which one the logged runs used is
decided by the logged errors.  Pinned by the Random123 known-answer vectors and by the value of
``split(PRNGKey(0))`` printed in JAX's own documentation (tests/test_threefry.py).
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def threefry2x32(key, x0, x1):
    """Threefry-2x32-20 of the counter words (x0, x1) under key = (k0, k1); arrays of 32-bit words (held in uint64)."""
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & _M32)
    x0 = (np.asarray(x0, dtype=np.uint64) + ks[0]) & _M32
    x1 = (np.asarray(x1, dtype=np.uint64) + ks[1]) & _M32
    for i in range(5):
        for r in _ROT[i % 2]:
            x0 = (x0 + x1) & _M32
            x1 = x0 ^ _rotl(x1, r)
        x0 = (x0 + ks[(i + 1) % 3]) & _M32
        x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & _M32
    return x0, x1


def _hash_counts(key, n):
    """``threefry_2x32(key, iota(n))`` of the original layout: the counters' first half is word 0, the second half word 1
    (an odd count is padded with one zero), outputs concatenated."""
    counts = np.arange(n + (n & 1), dtype=np.uint64)
    counts[n:] = 0
    h = counts.size // 2
    y0, y1 = threefry2x32(key, counts[:h], counts[h:])
    return np.concatenate([y0, y1])[:n]


def split(key, num=2, layout="original"):
    """``jax.random.split(key, num)`` -> (num, 2) key words."""
    if layout == "original":
        return _hash_counts(key, 2 * num).reshape(num, 2)
    if layout == "partitionable":
        y0, y1 = threefry2x32(key, np.zeros(num, np.uint64), np.arange(num, dtype=np.uint64))
        return np.stack([y0, y1], axis=1)
    raise ValueError("layout must be 'original' or 'partitionable'")


def random_bits32(key, n, layout="original"):
    """n 32-bit draws under ``key`` (``jax.random.bits(key, (n,), uint32)``)."""
    if layout == "original":
        return _hash_counts(key, n)
    if layout == "partitionable":
        y0, y1 = threefry2x32(key, np.zeros(n, np.uint64), np.arange(n, dtype=np.uint64))
        return y0 ^ y1
    raise ValueError("layout must be 'original' or 'partitionable'")


def choice_without_replacement(key, n, k, layout="original"):
    """``jax.random.choice(key, n, (k,), replace=False)`` for n < 2**21 (one sort round)."""
    if not 0 < k <= n < (1 << 21):
        raise ValueError("need 0 < k <= n < 2**21")
    sub = split(key, 2, layout)[1]
    order = np.argsort(random_bits32(sub, n, layout), kind="stable")
    return order[:k].astype(np.int32)


def reference_laplacian_idx(d, layout="original"):
    """The index set of models/GP.py:35 for spatial dimension d: ``choice(PRNGKey(0), d, (5,), replace=False)``."""
    return choice_without_replacement((0, 0), int(d), 5, layout)


def solver_key_words(q, n, key=(0, 0), quadrature=True):
    """Key words a solve on the reference's own random stream needs (``compat_rng="jax"``, SCASML_RNG_JAX_STREAM), and the solver's key
    state after it.  ``q[level][l]`` = quadrature nodes of sub-level l in a level-``level`` call (``scasml_plan.term[level][l].q``).

    The reference rebuilds ``split(PRNGKey(0), 1)[0]`` in every ``uz_solve`` call for its terminal draws (solvers/MLP.py:167-168, 178) -- word
    pair 0 -- and takes one sub-key per quadrature node from the solver's stateful key, ``self.key, subkey = random.split(self.key)``
    (:220), in call order: a level-n call consumes S(n) = sum_l q[n][l] (1 + S(l) + S(l-1)) of them, its children's included.  The
    full-history solvers draw everything from the first key (MLP_full_history.py:92-93): S = 0.
    Returns (uint32 array of shape (1 + S, 2), new key)."""
    def count(level):
        if level <= 0:
            return 0
        return sum(int(q[level][l]) * (1 + count(l) + count(l - 1)) for l in range(level))
    words = [split((0, 0), 1, "partitionable")[0]]
    key = (int(key[0]), int(key[1]))
    for _ in range(count(n) if quadrature else 0):
        pair = split(key, 2, "partitionable")
        key = (int(pair[0][0]), int(pair[0][1]))
        words.append(pair[1])
    return np.asarray(words, dtype=np.uint64).astype(np.uint32), key
