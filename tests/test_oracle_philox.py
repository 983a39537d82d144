"""Philox4x32-10 known-answer vectors (Random123 kat_vectors) and the normal generator's
statistical quality / bit-level determinism."""
import numpy as np

from oracle import philox

KAT = [  # (counter, key, expected) -- Random123 v1.14 examples/kat_vectors, philox4x32 10 rounds
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_philox_known_answers():
    for ctr, key, exp in KAT:
        got = tuple(int(v) for v in philox.philox4x32_10(*ctr, *key))
        assert got == exp


def test_normals_are_float32_deterministic_and_standard():
    a = philox.normals(7, 3, np.arange(4096), 11, 100)
    b = philox.normals(7, 3, np.arange(4096), 11, 100)
    assert a.dtype == np.float32 and a.shape == (4096, 100)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3
    assert abs((a.astype(np.float64) ** 4).mean() - 3) < 0.05
    assert np.abs(a).max() < 5.8                         # 24-bit uniforms: radius <= sqrt(2*24*ln2)
    # distinct sites / roots / streams / seeds decorrelate
    for other in (philox.normals(7, 3, np.arange(4096), 12, 100), philox.normals(7, 4, np.arange(4096), 11, 100),
                  philox.normals(8, 3, np.arange(4096), 11, 100), philox.normals(7, 3, np.arange(4096) + 4096, 11, 100)):
        assert abs(np.corrcoef(a.ravel(), other.ravel())[0, 1]) < 1e-2


def test_ragged_dimension_is_prefix_of_padded_quads():
    a = philox.normals(1, 0, np.arange(8), 5, 7)
    b = philox.normals(1, 0, np.arange(8), 5, 8)
    assert np.array_equal(a, b[:, :7])


def test_ln_and_sincos_accuracy():
    k = np.arange(1, 2 ** 24 + 1, 997, dtype=np.uint32)
    ref = np.log(k.astype(np.float64) * 2.0 ** -24)
    got = philox.ln_u24(k).astype(np.float64)
    assert np.abs(got - ref).max() < 5e-7 and (got <= 0).all()
    assert philox.ln_u24(np.array([2 ** 24], dtype=np.uint32))[0] == 0.0
    k = np.arange(0, 2 ** 24, 1013, dtype=np.uint32)
    c, s = philox.sincos_u24(k)
    assert np.abs(c.astype(np.float64) ** 2 + s.astype(np.float64) ** 2 - 1).max() < 3e-7


def test_uniform_tau_open_interval():
    u = philox.uniform_tau(0, 0, np.arange(200000), 3)
    assert u.dtype == np.float32 and u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 3e-3
