"""Error behaviour of the C ABI (include/pgi.h): nothing throws, misuse returns a negative pgi_error with a message."""
import ctypes as C

import numpy as np
import pytest

from pyposegraphbuilder import _lib as L

PGI_ERR_INVALID, PGI_ERR_TOO_LARGE = -1, -4


def test_null_context_is_rejected_everywhere():
    """CPU: every entry point taking a context refuses NULL before touching the device."""
    lib = L.load()
    z = C.c_void_p(0)
    calls = [
        lambda: lib.pgi_set_stream(z, z), lambda: lib.pgi_set_params(z, None), lambda: lib.pgi_synchronize(z),
        lambda: lib.pgi_estimate_pose_batch(z, None, z, z),
        lambda: lib.pgi_estimate_pose(z, z, 0, 0.0, z, 0, 0, 0, None, z),
        lambda: lib.pgi_estimate_pose_batch_host(z, z, z, z, z, z, z, z, z, 1, 0, 0, z, z),
        lambda: lib.pgi_desc_prepare_screen(z, z, 0, z, z),
        lambda: lib.pgi_desc_prepare(z, z, 0, z, z),
        lambda: lib.pgi_match_descriptors_batch(z, None, None, 1, 1, z, z, z, z),
        lambda: lib.pgi_build_correspondences(z, None, None, 1, 1, z, z, z, 0, 1.0, 0, z, z, z, z, z, z),
        lambda: lib.pgi_guided_match_batch(z, None, None, 1, z, 0, 1, z, z, z, z),
    ]
    for call in calls:
        assert call() < 0
        assert lib.pgi_last_error()
    lib.pgi_destroy(z)                                     # a no-op, not a crash
    assert lib.pgi_desc_padded(0) == 0 and lib.pgi_desc_padded(1) == 256 and lib.pgi_desc_padded(8000) == 8192
    p = L.Params()
    lib.pgi_default_params(C.byref(p))
    assert (p.confidence, p.max_iters, p.round_size, p.lo_iters, p.min_inliers) == (0.99, 1000, 32, 2, 20)


@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder import Engine
    e = Engine()
    yield e
    e.close()


@pytest.mark.gpu
def test_bad_arguments_on_a_live_context(eng):
    import torch
    lib, ctx = eng._lib, eng._ctx
    z = C.c_void_p(0)
    # missing batch pointers
    b = L.Batch()
    b.n_pairs = 1
    assert lib.pgi_estimate_pose_batch(ctx, C.byref(b), z, z) == PGI_ERR_INVALID
    buf = torch.zeros(1024, dtype=torch.uint8, device=eng.device)
    assert lib.pgi_estimate_pose_batch(ctx, C.byref(b), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr())) == PGI_ERR_INVALID
    assert b"batch pointers" in lib.pgi_last_error()
    # a descriptor view whose padding does not match pgi_desc_padded
    v = (L.DescView * 1)()
    v[0].d_desc_t, v[0].d_norm, v[0].n, v[0].n_pad = buf.data_ptr(), buf.data_ptr(), 10, 128
    out = C.c_void_p(buf.data_ptr())
    assert lib.pgi_match_descriptors_batch(ctx, v, v, 1, 4, out, out, out, out) == PGI_ERR_INVALID
    # more keypoints than the selection kernel supports
    assert lib.pgi_desc_prepare(ctx, out, 16385, out, out) == PGI_ERR_TOO_LARGE
    # a guess count without guesses on the single-pair seam
    corr = np.zeros((8, 4))
    e = L.Edge()
    m = np.zeros(8, np.uint8)
    assert lib.pgi_estimate_pose(ctx, corr.ctypes.data_as(C.c_void_p), 8, 1e-3, None, 1, 0, 0, C.byref(e),
                                 m.ctypes.data_as(C.c_void_p)) == PGI_ERR_INVALID
    # the context still works afterwards
    from pyposegraphbuilder import synthetic as S
    p = S.make_pair(1, 300)
    ok, edge, mask = eng.estimate_pose(np.stack([p["x1"], p["y1"], p["x2"], p["y2"]], 1), 7.5e-4)
    assert ok and int(mask.sum()) == edge.n_inl


@pytest.mark.gpu
def test_contexts_do_not_leak(eng):
    """Create/destroy cycles with all workspaces exercised leave device memory where it was."""
    import torch
    from pyposegraphbuilder import Engine, synthetic as S
    rng = np.random.default_rng(3)
    A, B, _ = S.make_descriptors(rng, 600, 700)

    def cycle():
        e = Engine()
        im = [e.prepare_descriptors(A), e.prepare_descriptors(B)]
        e.match_descriptors_batch(im, [(0, 1), (1, 0)])
        p = S.make_pair(2, 500)
        e.estimate_pose(np.stack([p["x1"], p["y1"], p["x2"], p["y2"]], 1), 7.5e-4)
        del im
        e.close()

    cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)
