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
    rng = _lib.Rng(seed, stream, root0, 0, 1)
    _lib.check(lib.scasml_debug_normals(rng, site, d, B, _lib.ptr(out), _lib.stream_ptr()), "debug_normals")
    got = out.cpu().numpy()
    want = philox.normals(seed, stream, np.arange(root0, root0 + B), site, d)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
