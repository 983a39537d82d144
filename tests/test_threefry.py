"""The generator behind the reference's Hutchinson index set (models/GP.py:35), pinned without JAX: the Random123
known-answer vectors of Threefry-2x32-20 (the ones JAX's own test-suite uses) and the value of
``split(PRNGKey(0))`` that JAX's documentation prints."""
import numpy as np

from scasml_gp_amd import threefry as tf


def test_threefry2x32_known_answers():
    kats = [((0x0, 0x0), (0x0, 0x0), (0x6B200159, 0x99BA4EFE)),
            ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
            ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]
    for key, ctr, want in kats:
        y0, y1 = tf.threefry2x32(key, [ctr[0]], [ctr[1]])
        assert (int(y0[0]), int(y1[0])) == want


def test_split_of_key_zero_is_the_documented_pair():
    got = tf.split((0, 0), 2, "original")
    assert got.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]


def test_reference_index_sets_are_valid_and_frozen():
    # frozen from this implementation once the two pins above held; a change here is a change of the compat surrogate
    frozen = {("original", 20): [18, 17, 15, 9, 11], ("original", 40): [2, 14, 21, 37, 39], ("original", 60): [53, 37, 5, 52, 10],
              ("original", 80): [11, 23, 2, 17, 6], ("original", 100): [29, 15, 35, 8, 40],
              ("partitionable", 20): [0, 1, 19, 8, 12], ("partitionable", 40): [0, 36, 1, 19, 31],
              ("partitionable", 60): [0, 36, 1, 40, 19], ("partitionable", 80): [0, 36, 1, 40, 19]}
    for (layout, d), want in frozen.items():
        idx = tf.reference_laplacian_idx(d, layout)
        assert idx.tolist() == want
        assert len(set(idx.tolist())) == 5 and idx.min() >= 0 and idx.max() < d


def test_odd_counts_are_padded_like_jax():
    a = tf.random_bits32((1, 2), 7, "original")
    b = tf.random_bits32((1, 2), 8, "original")
    assert a.shape == (7,) and not np.array_equal(a, b[:7])      # halves move with the size: the layouts are size-dependent
    y0, y1 = tf.threefry2x32((1, 2), [0, 1, 2, 3], [4, 5, 6, 0])
    assert np.array_equal(a, np.concatenate([y0, y1])[:7])


def test_solver_key_words_follow_the_reference_call_order():
    """The host side of SCASML_RNG_JAX_STREAM (scasml_gp_amd/threefry.py::solver_key_words) against the replay of the reference's recursion
    (oracle/replay.py), which consumes its sub-keys in the reference's call order: same sub-keys, same key state after one and two solves."""
    from oracle import jax_random as jr
    from oracle.equation import GradDependentNonlinear
    from oracle.replay import ReplayMLP
    from oracle.tables import approx_parameters
    from scasml_gp_amd.threefry import solver_key_words
    rho = n = 2
    _, _, Q, _, _ = approx_parameters(rho, 0.5)
    q = [[int(Q[rho - 1, level - l - 1]) for l in range(level)] for level in range(n + 1)]       # MLP.py:211
    words, key = solver_key_words(q, n)
    assert words.shape == (16, 2) and words.dtype == np.uint32
    assert np.array_equal(words[0], jr.split(jr.prng_key(0), 1)[0].astype(np.uint32))
    rep = ReplayMLP(GradDependentNonlinear(6))
    drawn = []
    take = rep._next_subkey
    rep._next_subkey = lambda: drawn.append(take()) or drawn[-1]
    x = np.zeros((2, 6), dtype=np.float16)
    rep.uz_solve(n, rho, x)
    assert len(drawn) == 15 and np.array_equal(np.asarray(drawn).astype(np.uint32), words[1:])
    assert tuple(int(v) for v in rep.key) == key
    words2, key2 = solver_key_words(q, n, key)
    rep.uz_solve(n, rho, x)
    assert np.array_equal(np.asarray(drawn[15:]).astype(np.uint32), words2[1:]) and tuple(int(v) for v in rep.key) == key2
    fh, same = solver_key_words(q, n, key, quadrature=False)
    assert fh.shape == (1, 2) and same == key and np.array_equal(fh[0], words[0])
