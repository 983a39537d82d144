"""Device Philox + normal transform must equal the oracle's NumPy statement bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,site,seed,stream,root0", [(100, 0, 0, 0, 0), (7, 12345, 0xDEADBEEFCAFE, 9, 1 << 20), (254, 2 ** 32 - 1, 1, 2 ** 32 - 1, 77)])
def test_normals_bit_identical(d, site, seed, stream, root0):
    import torch
    from oracle import philox
    from scasml_gp_amd import _lib
    lib = _lib.load()
    B = 4096
    out = torch.empty((B, d), dtype=torch.float32, device="cuda")
    rng = _lib.Rng(seed, stream, root0, 0, 1, 0, 0)
    _lib.check(lib.scasml_debug_normals(rng, site, d, B, _lib.ptr(out), _lib.stream_ptr()), "debug_normals")
    got = out.cpu().numpy()
    want = philox.normals(seed, stream, np.arange(root0, root0 + B), site, d)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_normal_transform_is_bit_identical_on_every_input():
    """A normal is a function of the top 24 bits of its Philox word: the device transform is compared with the NumPy statement
    on ALL 2^24 inputs, so every normal the device can produce has the oracle's bits.  The library's table (the committed
    constant) must be the oracle's restatement of its definition."""
    import torch
    from oracle import philox
    from scasml_gp_amd import _lib
    lib = _lib.load()
    rows = lib.scasml_normal_table_rows()
    table = np.empty((rows, 4), dtype=np.float32)
    _lib.check(lib.scasml_normal_table(table.ctypes.data_as(C.c_void_p), rows), "normal_table")
    assert np.array_equal(table.view(np.uint32), philox.normal_table().view(np.uint32))
    n = 1 << 24
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(lib.scasml_debug_transform(0, n, _lib.ptr(out), _lib.stream_ptr()), "debug_transform")
    want = philox.icdf_normal(np.arange(n, dtype=np.uint32) << np.uint32(8))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
