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
    """Box-Muller here is separable: the radius depends on one 24-bit word, (cos, sin) on another, and a normal is
    one IEEE product of the two.  Both halves are compared with the NumPy statement on their WHOLE domains
    (2^24 inputs each), so every normal the device can produce has the oracle's bits."""
    import torch
    from oracle import philox
    from scasml_gp_amd import _lib
    lib = _lib.load()
    n = 1 << 24
    rad = torch.empty(n, dtype=torch.float32, device="cuda")
    cs = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    _lib.check(lib.scasml_debug_transform(0, n, _lib.ptr(rad), _lib.ptr(cs), _lib.stream_ptr()), "debug_transform")
    k = np.arange(n, dtype=np.uint32)
    t = np.float32(-2.0) * philox.ln_u24(k + np.uint32(1))
    t = np.where(t < 0, np.float32(0.0), t)
    want_rad = np.sqrt(t).astype(np.float32)
    want_c, want_s = philox.sincos_u24(k)
    got_rad, got_cs = rad.cpu().numpy(), cs.cpu().numpy()
    assert np.array_equal(got_rad.view(np.uint32), want_rad.view(np.uint32))
    assert np.array_equal(got_cs[:, 0].view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(got_cs[:, 1].view(np.uint32), want_s.view(np.uint32))
