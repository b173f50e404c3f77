"""Error behaviour of the C ABI (include/pgi.h): nothing throws, misuse returns a negative pgi_error with a message."""
import ctypes as C

import numpy as np
import pytest

from pyposegraphbuilder import _lib as L

PGI_ERR_INVALID, PGI_ERR_TOO_LARGE, PGI_ERR_COMM = -1, -4, -5


def test_null_context_is_rejected_everywhere():
    """CPU: every entry point taking a context refuses NULL before touching the device."""
    lib = L.load()
    z = C.c_void_p(0)
    calls = [
        lambda: lib.pgi_set_stream(z, z), lambda: lib.pgi_set_params(z, None), lambda: lib.pgi_synchronize(z),
        lambda: lib.pgi_estimate_pose_batch(z, None, z, z),
        lambda: lib.pgi_estimate_pose(z, z, 0, 0.0, z, 0, 0, 0, 0, None, z),
        lambda: lib.pgi_estimate_pose_batch_host(z, z, z, z, z, z, z, z, z, 1, 0, 0, z, z),
        lambda: lib.pgi_desc_prepare_screen(z, z, 0, z, z),
        lambda: lib.pgi_desc_prepare(z, z, 0, z, z),
        lambda: lib.pgi_match_descriptors_batch(z, None, None, 1, 1, z, z, z, z),
        lambda: lib.pgi_build_correspondences(z, None, None, 1, 1, z, z, z, 0, 1.0, 0, z, z, z, z, z, z),
        lambda: lib.pgi_guided_match_batch(z, None, None, 1, z, 0, 0, 1, z, z, z, z),
        lambda: lib.pgi_get_params(z, None), lambda: lib.pgi_comm_destroy(z), lambda: lib.pgi_comm_info(z, None, None, None),
        lambda: lib.pgi_comm_init_rccl(z, 2, 0, z), lambda: lib.pgi_comm_init_host(z, 2, 0, L.ALLGATHERV_FN(0), z),
        lambda: lib.pgi_allgather_edges(z, z, z, z), lambda: lib.pgi_allgatherv(z, z, z, z),
        lambda: lib.pgi_rotation_average_edges(z, z, z, z, z, 0, 1, None, z, z, z),
        lambda: lib.pgi_comm_unique_id(z),
        lambda: lib.pgi_score_pose_f64_host(z, z, 4, z, 1.0, 5, None, z),
        lambda: lib.pgi_pose_from_essential_host(z, z, z, 4, z, z, z, None, None),
        lambda: lib.pgi_screen_guesses(z, z, 5, z, 4),
        lambda: lib.pgi_tracklets_add_batch(z, None, 1), lambda: lib.pgi_tracklets_get_batch(z, z, z, 1, 1, 2, z, z, z),
        lambda: lib.pgi_tracklets_info(z, None, None, None), lambda: lib.pgi_tracklets_track(z, 0, z, 0, None),
    ]
    for call in calls:
        assert call() < 0
        assert lib.pgi_last_error()
    assert not lib.pgi_tracklets_create(z, 4) and lib.pgi_last_error()
    lib.pgi_tracklets_destroy(z)                           # no-ops, not crashes
    lib.pgi_destroy(z)
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
    assert lib.pgi_estimate_pose(ctx, corr.ctypes.data_as(C.c_void_p), 8, 1e-3, None, 1, 0, 0, 0, C.byref(e),
                                 m.ctypes.data_as(C.c_void_p)) == PGI_ERR_INVALID
    # communicator misuse: rank outside the world; no callback
    assert lib.pgi_comm_init_host(ctx, 2, 2, L.ALLGATHERV_FN(lambda *a: 0), None) == PGI_ERR_INVALID
    w, r, k = C.c_uint32(9), C.c_uint32(9), C.c_uint32(9)
    assert lib.pgi_comm_info(ctx, C.byref(w), C.byref(r), C.byref(k)) == 0 and (w.value, r.value, k.value) == (1, 0, 0)
    # guided matching: an output stride that cannot hold what a pair may return is refused, not silently clamped
    fv = (L.FeatureView * 1)()
    fv[0].d_xy, fv[0].d_desc, fv[0].n = buf.data_ptr(), buf.data_ptr(), 100
    fv[0].fx = fv[0].fy = 1000.0
    fv[0].cx, fv[0].cy, fv[0].width, fv[0].height = 800.0, 600.0, 1600.0, 1200.0
    pose = np.r_[np.eye(3).ravel(), [1.0, 0.0, 0.0]]
    assert lib.pgi_guided_match_batch(ctx, fv, fv, 1, pose.ctypes.data_as(C.c_void_p), 45, 0, 50, out, out, out, out) == PGI_ERR_INVALID
    assert lib.pgi_guided_match_batch(ctx, fv, fv, 1, pose.ctypes.data_as(C.c_void_p), 45, 100, 50, out, out, out, out) == PGI_ERR_INVALID
    assert b"out_stride" in lib.pgi_last_error()
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


@pytest.mark.gpu
def test_single_pair_seam_is_reentrant_with_per_call_min_inliers(eng):
    """The reference calls estimatePose from kCoreNumber OpenMP threads (pose_graph_builder.h:391-392), each with its own
    kMinimumInlierNumber_ argument (:155): 24 concurrent callers with two different minimum-inlier values must each get
    the single-threaded answer, and a per-call value must not leak into the context's parameters."""
    import threading
    from pyposegraphbuilder import synthetic as S
    pairs = [S.make_pair(100 + i, 200 + 37 * i, inlier_ratio=0.5) for i in range(24)]
    corr = [np.stack([p["x1"], p["y1"], p["x2"], p["y2"]], 1).astype(np.float64) for p in pairs]
    # a minimum nobody reaches for odd callers: they must report failure while even callers succeed
    mins = [0 if i % 2 == 0 else 100000 for i in range(24)]
    ref = [eng.estimate_pose(corr[i], 7.5e-4, seed=5, pair_id=i, min_inliers=mins[i]) for i in range(24)]
    assert all(ref[i][0] == (i % 2 == 0) for i in range(24))
    out = [None] * 24

    def work(i):
        for _ in range(3):
            out[i] = eng.estimate_pose(corr[i], 7.5e-4, seed=5, pair_id=i, min_inliers=mins[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(24):
        assert out[i][0] == ref[i][0] and np.array_equal(out[i][2], ref[i][2])
        assert bytes(out[i][1]) == bytes(ref[i][1])
    p = L.Params()
    assert eng._lib.pgi_get_params(eng._ctx, C.byref(p)) == 0 and p.min_inliers == 20


@pytest.mark.gpu
def test_stream_switch_orders_context_scratch(eng):
    """pgi_set_stream must order the new stream after the old one: calls that share context-owned scratch, issued under
    different torch streams without any host synchronisation, give the same result as the serial sequence."""
    import torch
    from pyposegraphbuilder import synthetic as S
    rng = np.random.default_rng(11)
    A, B, _ = S.make_descriptors(rng, 3000, 3100)
    A2, B2, _ = S.make_descriptors(rng, 2900, 3050)
    im = [eng.prepare_descriptors(x) for x in (A, B, A2, B2)]
    torch.cuda.synchronize()
    ref1 = [t.clone() for t in eng.match_descriptors_batch(im, [(0, 1)] * 6, raw=True)]
    ref2 = [t.clone() for t in eng.match_descriptors_batch(im, [(2, 3)] * 6, raw=True)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            g1 = eng.match_descriptors_batch(im, [(0, 1)] * 6, raw=True)
        with torch.cuda.stream(s2):   # reuses the matching workspace while s1's kernels may still be running
            g2 = eng.match_descriptors_batch(im, [(2, 3)] * 6, raw=True)
        torch.cuda.synchronize()
        n1, n2 = int(ref1[3][0]), int(ref2[3][0])
        assert torch.equal(g1[3], ref1[3]) and torch.equal(g2[3], ref2[3])
        assert torch.equal(g1[0][:, :n1], ref1[0][:, :n1]) and torch.equal(g2[0][:, :n2], ref2[0][:, :n2])
        assert torch.equal(g1[2][:, :n1], ref1[2][:, :n1]) and torch.equal(g2[2][:, :n2], ref2[2][:, :n2])


@pytest.mark.gpu
def test_rccl_communicator_of_one_rank(eng):
    """The RCCL transport end to end on the one GPU there is: librccl is opened at run time, a unique id is created,
    a one-rank communicator is initialised and pgi_allgather_edges runs ncclAllGather on the engine's stream."""
    import torch
    lib, ctx = eng._lib, eng._ctx
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    L.check(lib.pgi_comm_unique_id(ident))
    assert any(ident)
    L.check(lib.pgi_comm_init_rccl(ctx, 1, 0, ident))
    try:
        assert eng.comm_info() == (1, 0, "rccl")
        rec = np.zeros(37, L.EDGE_DTYPE)
        rec["n_inl"] = np.arange(37)
        rec["R"] = np.arange(37)[:, None] * 0.5 + np.arange(9)[None, :]
        local = torch.from_numpy(rec.view(np.uint8).reshape(37, 200).copy()).to(eng.device)
        out = eng.allgather_edges(local, [37])
        torch.cuda.synchronize()
        assert torch.equal(out, local)
    finally:
        L.check(lib.pgi_comm_destroy(ctx))
    assert eng.comm_info() == (1, 0, "none")


@pytest.mark.gpu
def test_host_scheduler_entry_validates_caller_data(capfd):
    """pgih_run_pairs (include/pgi_host.h) sizes host tables from the caller's view ids: ids beyond n_views (or, with n_views = 0,
    beyond PGIH_MAX_VIEWS), decreasing offsets and null rows are refused with a message, before anything is allocated from
    them; a well-formed call still works afterwards, takes a seed, and says so when most accepted guesses are quirk-only."""
    from pyposegraphbuilder import PoseGraphBuilder
    import scene_drivers as SC
    b = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.4, "", "", "", "", True, True, True)
    try:
        lib, h = b._host()
        src, dst = np.array([0, 1], np.uint32), np.array([1, 2], np.uint32)
        sim, thr = np.array([0.9, 0.8]), np.array([7.5e-4, 7.5e-4])
        off = np.array([0, 60, 120], np.uint64)
        corr = np.zeros((120, 4))
        edges = np.zeros(4, [("src", "<u4"), ("dst", "<u4"), ("score", "<f8"), ("R", "<f8", 9), ("t", "<f8", 3)])
        n_edges = C.c_uint32(0)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)

        def call(n_views, src_, dst_, off_, corr_ptr):
            return lib.pgih_run_pairs(h, n_views, 2, ptr(src_), ptr(dst_), ptr(sim), ptr(thr), ptr(off_), corr_ptr, 64, 0, ptr(edges), 4,
                                      C.byref(n_edges), None)
        assert call(2, src, dst, off, ptr(corr)) < 0 and b"out of range" in lib.pgih_last_error()          # id 2 with n_views = 2
        big = np.array([1, 0xFFFFFFFF], np.uint32)
        assert call(0, src, big, off, ptr(corr)) < 0 and b"PGIH_MAX_VIEWS" in lib.pgih_last_error()       # derived count unbounded
        assert call(3, src, dst, np.array([0, 60, 30], np.uint64), ptr(corr)) < 0 and b"offsets" in lib.pgih_last_error()
        assert call(3, src, dst, off, None) < 0 and b"null" in lib.pgih_last_error()
        assert call(3, src, dst, off, ptr(corr)) == 0 and n_edges.value == 0                                 # all-zero rows: no edge, no crash
        # a real run through the Python class: seed accepted, rotationGuided keyword-only and not sticky, warning on quirk-only guesses
        g, wave = SC.make_scene("v340_thin")
        bt, simv = g["batch"], SC.pair_similarity(g)
        pairs = []
        for e, (i, j) in enumerate(g["pairs"]):
            a, z = int(bt["offsets"][e]), int(bt["offsets"][e + 1])
            pairs.append(dict(src=int(i), dst=int(j), similarity=float(simv[e]), threshold=7.5e-4,
                              correspondences=np.stack([bt["x1"][a:z], bt["y1"][a:z], bt["x2"][a:z], bt["y2"][a:z]], 1)))
        with pytest.raises(TypeError):
            b.run(pairs, wave, 0, True)                      # rotationGuided cannot be passed positionally any more
        capfd.readouterr()
        g0 = b.run(pairs, waveSize=wave, seed=0, numViews=340)
        err = capfd.readouterr().err
        assert b.statistics["quirk_only_guesses"] * 20 > b.statistics["poses_from_guess"] and "setRotationGuidedGuesses" in err
        g7 = b.run(pairs, waveSize=wave, seed=7)
        assert set(g0) != set(g7) or any(not np.array_equal(g0[k]["R"], g7[k]["R"]) for k in g0 if k in g7)  # another seed, other draws
        gg = b.run(pairs, waveSize=wave, rotationGuided=True)
        assert b.statistics["quirk_only_guesses"] == 0
        again = b.run(pairs, waveSize=wave, seed=0)          # the switch did not stick, the run is reproducible
        assert set(again) == set(g0) and all(np.array_equal(again[k]["R"], g0[k]["R"]) and again[k]["score"] == g0[k]["score"] for k in g0)
        assert len(gg) > 0
    finally:
        b.close()
