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
    assert np.abs(a).max() < 5.43                        # 24-bit uniforms: |Phi^-1(2^-25)| = 5.42
    # distinct sites / roots / streams / seeds decorrelate
    for other in (philox.normals(7, 3, np.arange(4096), 12, 100), philox.normals(7, 4, np.arange(4096), 11, 100),
                  philox.normals(8, 3, np.arange(4096), 11, 100), philox.normals(7, 3, np.arange(4096) + 4096, 11, 100)):
        assert abs(np.corrcoef(a.ravel(), other.ravel())[0, 1]) < 1e-2


def test_ragged_dimension_is_prefix_of_padded_quads():
    a = philox.normals(1, 0, np.arange(8), 5, 7)
    b = philox.normals(1, 0, np.arange(8), 5, 8)
    assert np.array_equal(a, b[:, :7])


def test_normal_transform_on_its_whole_domain():
    """Every one of the 2^24 inputs: within 5e-7 (one binary32 ulp at |x| > 4) of scipy's inverse CDF of the input's uniform, odd about u = 1/2, and
    monotone to rounding (a decrease between neighbouring inputs is at most one rounding of the cubic's value)."""
    from scipy.special import ndtri
    k = np.arange(1 << 24, dtype=np.uint32)
    x = philox.icdf_normal(k << np.uint32(8))
    assert x.dtype == np.float32
    u = (k.astype(np.float64) + 0.5) * 2.0 ** -24
    assert np.abs(x.astype(np.float64) - ndtri(u)).max() < 5e-7
    assert np.array_equal(x[: 1 << 23], -x[1 << 23:][::-1])
    assert np.diff(x.astype(np.float64)).min() > -5e-7 and x[0] < -5.41 and x[-1] > 5.41
    # the low 8 bits of a Philox word do not matter
    assert np.array_equal(philox.icdf_normal((k[::4099] << np.uint32(8)) | np.uint32(0xFF)), x[::4099])


def test_committed_table_is_the_oracles_table():
    """The product's constant (scasml_gp_amd/csrc/normal_table.inc, hex floats) against the oracle's restatement of the definition."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scasml_gp_amd", "csrc", "normal_table.inc")
    rows = [[float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", line)] for line in open(path) if line.startswith("{")]
    got = np.asarray(rows, dtype=np.float64)
    want = philox.normal_table()
    assert got.shape == (768, 4) == want.shape
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.view(np.uint32)) and np.array_equal(got, want.astype(np.float64))


def test_uniform_tau_open_interval():
    u = philox.uniform_tau(0, 0, np.arange(200000), 3)
    assert u.dtype == np.float32 and u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 3e-3
