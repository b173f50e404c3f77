"""Engine: the batched entry points of include/pgi.h over torch device tensors.

torch is used for HBM allocation and the stream only; all compute is in libpgi.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    def __init__(self, device=None, **params):
        lib = L.load()
        if not torch.cuda.is_available():
            raise L.PgiError("no HIP device visible: the pose engine has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.params = L.default_params(**params)
        self._ctx = lib.pgi_create(self.device.index, C.byref(self.params))
        if not self._ctx:
            raise L.PgiError("pgi_create failed: " + L.last_error())
        self._lib = lib

    def close(self):
        if getattr(self, "_ctx", None):
            for child in list(getattr(self, "_children", ())):  # stores that live on this context go first
                child.close()
            self._lib.pgi_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.params, k, v)
        L.check(self._lib.pgi_set_params(self._ctx, C.byref(self.params)))

    def _bind_stream(self):
        L.check(self._lib.pgi_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def synchronize(self):
        L.check(self._lib.pgi_synchronize(self._ctx))

    # ---- batches -------------------------------------------------------------------------
    def upload(self, x1, y1, x2, y2, offsets, thr, guesses=None, has_guess=None, seed=0, pair_id_base=0):
        """Host (numpy) flattened SoA -> a device-resident batch dict."""
        dev = self.device
        off = np.ascontiguousarray(offsets, np.uint64)
        P = len(off) - 1
        b = dict(
            x1=torch.from_numpy(np.ascontiguousarray(x1, np.float32)).to(dev),
            y1=torch.from_numpy(np.ascontiguousarray(y1, np.float32)).to(dev),
            x2=torch.from_numpy(np.ascontiguousarray(x2, np.float32)).to(dev),
            y2=torch.from_numpy(np.ascontiguousarray(y2, np.float32)).to(dev),
            offsets=torch.from_numpy(off.view(np.int64)).to(dev),
            thr=torch.from_numpy(np.array(np.broadcast_to(thr, (P,)), np.float64)).to(dev),
            guesses=None, has_guess=None, n_pairs=P,
            max_corr=int(np.diff(off.astype(np.int64)).max()) if P else 0,
            seed=int(seed), pair_id_base=int(pair_id_base))
        if guesses is not None:
            b["guesses"] = torch.from_numpy(np.ascontiguousarray(guesses, np.float64).reshape(P, 12)).to(dev)
            hg = np.ones(P, np.uint8) if has_guess is None else np.ascontiguousarray(has_guess, np.uint8)
            b["has_guess"] = torch.from_numpy(hg).to(dev)
        return b

    def _batch_struct(self, b):
        s = L.Batch()
        s.d_x1, s.d_y1, s.d_x2, s.d_y2 = (b[k].data_ptr() for k in ("x1", "y1", "x2", "y2"))
        s.d_offsets = b["offsets"].data_ptr()
        s.d_thr = b["thr"].data_ptr()
        s.d_guess_Rt = b["guesses"].data_ptr() if b.get("guesses") is not None else None
        s.d_has_guess = b["has_guess"].data_ptr() if b.get("has_guess") is not None else None
        s.n_pairs, s.max_corr = b["n_pairs"], b["max_corr"]
        s.pair_id_base, s.seed = b["pair_id_base"], b["seed"]
        return s

    def estimate_pose_batch(self, b, edges=None, masks=None):
        """Enqueues estimatePose for every pair; returns (edges uint8[P,200], masks uint8[rows]) on device."""
        P, rows = b["n_pairs"], b["x1"].numel()
        if edges is None:
            edges = torch.empty((P, L.EDGE_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        if masks is None:
            masks = torch.empty(max(rows, 1), dtype=torch.uint8, device=self.device)
        self._bind_stream()
        s = self._batch_struct(b)
        L.check(self._lib.pgi_estimate_pose_batch(self._ctx, C.byref(s), _ptr(edges), _ptr(masks)))
        return edges, masks[:rows]

    def estimate_pose_batch_host(self, x1, y1, x2, y2, offsets, thr, guesses=None, has_guess=None, seed=0, pair_id_base=0,
                                 out=None):
        """Host (numpy) SoA in, (edges structured array, masks) out; copies are pipelined against the kernels.
        out = (edges, masks): caller-provided (e.g. pinned) result arrays."""
        f = lambda a: np.ascontiguousarray(a, np.float32)
        x1, y1, x2, y2 = f(x1), f(y1), f(x2), f(y2)
        off = np.ascontiguousarray(offsets, np.uint64)
        P = len(off) - 1
        thr = np.array(np.broadcast_to(thr, (P,)), np.float64)
        g = hg = None
        if guesses is not None:
            g = np.ascontiguousarray(guesses, np.float64).reshape(P, 12)
            hg = np.ones(P, np.uint8) if has_guess is None else np.ascontiguousarray(has_guess, np.uint8)
        if out is None:
            edges = np.zeros(P, L.EDGE_DTYPE)
            masks = np.zeros(max(int(off[-1] - off[0]), 1), np.uint8)
        else:
            edges, masks = out
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        L.check(self._lib.pgi_estimate_pose_batch_host(self._ctx, p(x1), p(y1), p(x2), p(y2), p(off), p(thr), p(g), p(hg), P,
                                                       int(pair_id_base), int(seed), p(edges), p(masks)))
        return edges, masks[:int(off[-1] - off[0])]

    def pin(self, *arrays):
        """Page-locks numpy arrays (pgi_host_register) so estimate_pose_batch_host moves them by asynchronous DMA; returns
        them.  Call unpin() on the same arrays before they are freed."""
        for a in arrays:
            if a.nbytes:
                L.check(self._lib.pgi_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes))
        return arrays

    def unpin(self, *arrays):
        for a in arrays:
            if a.nbytes:
                L.check(self._lib.pgi_host_unregister(a.ctypes.data_as(C.c_void_p)))

    @staticmethod
    def edges_to_numpy(edges):
        return edges.cpu().numpy().view(L.EDGE_DTYPE).reshape(-1)

    def score_pose_batch(self, b, E, tau2, want_masks=True):
        P, rows = b["n_pairs"], b["x1"].numel()
        E = torch.as_tensor(np.ascontiguousarray(E, np.float64).reshape(P, 9)).to(self.device)
        tau2 = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(tau2, (P,)), np.float64)).to(self.device)
        counts = torch.empty(P, dtype=torch.int32, device=self.device)
        masks = torch.empty(max(rows, 1), dtype=torch.uint8, device=self.device) if want_masks else None
        self._bind_stream()
        s = self._batch_struct(b)
        L.check(self._lib.pgi_score_pose_batch(self._ctx, C.byref(s), _ptr(E), _ptr(tau2), _ptr(counts), _ptr(masks)))
        return counts, (masks[:rows] if want_masks else None)

    def score_pose_f64(self, corr_aos, offsets, E, tau2):
        """Reference-layout rows (n x 4 f64, cv::Mat N x 4 CV_64F) and arithmetic."""
        off = np.ascontiguousarray(offsets, np.uint64)
        P = len(off) - 1
        dev = self.device
        c = torch.from_numpy(np.ascontiguousarray(corr_aos, np.float64)).to(dev)
        o = torch.from_numpy(off.view(np.int64)).to(dev)
        E = torch.from_numpy(np.ascontiguousarray(E, np.float64).reshape(P, 9)).to(dev)
        t2 = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tau2, (P,)), np.float64)).to(dev)
        counts = torch.empty(P, dtype=torch.int32, device=dev)
        masks = torch.empty(max(len(corr_aos), 1), dtype=torch.uint8, device=dev)
        self._bind_stream()
        L.check(self._lib.pgi_score_pose_f64(self._ctx, _ptr(c), _ptr(o), P, _ptr(E), _ptr(t2), _ptr(counts),
                                             _ptr(masks)))
        return counts.cpu().numpy().astype(np.uint32), masks[:len(corr_aos)].cpu().numpy()

    def score_pose_host(self, corr_aos, E, tau2, early_exit_at=0, want_mask=True):
        """One pair, host pointers, re-entrant (pgi_score_pose_f64_host): getInliers (early_exit_at = 0, mask) /
        InTraversalPoseTester::test (early_exit_at = kMinimumInlierNumber, no mask) -> (reached, count, mask or None)."""
        c = np.ascontiguousarray(corr_aos, np.float64).reshape(-1, 4)
        Ed = np.ascontiguousarray(E, np.float64).reshape(9)
        n = len(c)
        mask = np.zeros(n, np.uint8) if want_mask else None
        cnt = C.c_uint32(0)
        rc = self._lib.pgi_score_pose_f64_host(self._ctx, c.ctypes.data_as(C.c_void_p), C.c_uint32(n), Ed.ctypes.data_as(C.c_void_p),
                                               C.c_double(tau2), C.c_uint32(early_exit_at), C.byref(cnt),
                                               mask.ctypes.data_as(C.c_void_p) if want_mask and n else None)
        if rc < 0:
            L.check(rc)
        return rc == 1, int(cnt.value), mask

    def pose_from_essential_host(self, E, corr_aos, mask=None):
        """pose::getPoseFromEssentialMatrix (pose_utils.h:172-252) for ONE pair with host pointers, re-entrant
        (pgi_pose_from_essential_host): -> (R[3,3], t[3], votes, cand).  mask None: every row votes (the reference)."""
        c = np.ascontiguousarray(corr_aos, np.float64).reshape(-1, 4)
        Ed = np.ascontiguousarray(E, np.float64).reshape(9)
        n = len(c)
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        if m is not None and len(m) != n:
            raise ValueError("mask must have one byte per row")
        R, t = np.zeros(9), np.zeros(3)
        votes, cand = C.c_uint32(0), C.c_uint32(0)
        L.check(self._lib.pgi_pose_from_essential_host(self._ctx, Ed.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p) if n else None,
                                                       C.c_uint32(n), m.ctypes.data_as(C.c_void_p) if m is not None and n else None,
                                                       R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.byref(votes), C.byref(cand)))
        return R.reshape(3, 3), t, int(votes.value), int(cand.value)

    def decompose_batch(self, b, E, masks=None):
        P = b["n_pairs"]
        E = torch.as_tensor(np.ascontiguousarray(E, np.float64).reshape(P, 9)).to(self.device)
        edges = torch.zeros((P, L.EDGE_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        self._bind_stream()
        s = self._batch_struct(b)
        L.check(self._lib.pgi_decompose_batch(self._ctx, C.byref(s), _ptr(E), _ptr(masks), _ptr(edges)))
        return edges

    def five_point_batch(self, pts, debug=False):
        """pts: [S,5,4] float32 -> (models [S,10,9] f32, counts [S], dbg [S,358] f64 or None)."""
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 5, 4)
        S = len(pts)
        d_pts = torch.from_numpy(pts).to(self.device)
        models = torch.zeros((S, 10, 9), dtype=torch.float32, device=self.device)
        counts = torch.zeros(S, dtype=torch.int32, device=self.device)
        dbg = torch.zeros((S, L.DBG_DOUBLES), dtype=torch.float64, device=self.device) if debug else None
        self._bind_stream()
        L.check(self._lib.pgi_five_point_batch(self._ctx, _ptr(d_pts), S, _ptr(models), _ptr(counts), _ptr(dbg)))
        return models.cpu().numpy(), counts.cpu().numpy(), (dbg.cpu().numpy() if debug else None)

    # ---- rotation averaging -----------------------------------------------------------------------
    def rotation_average(self, src, dst, R_rel, weight, n_views, **kw):
        """L1 + IRLS rotation averaging over edges (src, dst, R_dst_src, weight) -> (R[n_views,3,3], iters)."""
        E = len(src)
        edges = np.zeros(E, L.ROT_EDGE_DTYPE)
        edges["src"], edges["dst"] = src, dst
        edges["R"] = np.asarray(R_rel, np.float64).reshape(E, 9)
        edges["weight"] = weight
        prm = L.RotAvgParams()
        self._lib.pgi_default_rotavg_params(C.byref(prm))
        for k, v in kw.items():
            if not hasattr(prm, k):
                raise TypeError("unknown parameter %r" % k)
            setattr(prm, k, v)
        R = np.zeros((n_views, 3, 3))
        iters = C.c_uint32(0)
        self._bind_stream()
        L.check(self._lib.pgi_rotation_average(self._ctx, edges.ctypes.data_as(C.c_void_p), E, n_views, C.byref(prm),
                                               R.ctypes.data_as(C.c_void_p), C.byref(iters)))
        return R, iters.value

    def rotation_average_edges(self, edges, src, dst, rows, n_views, **kw):
        """The same solve fed from the device-resident edge table (uint8 [P, 200], e.g. the output of allgather_edges):
        only status / n_inl and the spanning forest's rotations travel to the host.  weight = n_inl / rows.
        -> (R[n_views,3,3], iters, edges_used)"""
        P = int(edges.shape[0])
        src = np.ascontiguousarray(src, np.uint32)
        dst = np.ascontiguousarray(dst, np.uint32)
        rows = None if rows is None else np.ascontiguousarray(rows, np.uint32)
        prm = L.RotAvgParams()
        self._lib.pgi_default_rotavg_params(C.byref(prm))
        for k, v in kw.items():
            if not hasattr(prm, k):
                raise TypeError("unknown parameter %r" % k)
            setattr(prm, k, v)
        R = np.zeros((n_views, 3, 3))
        iters, used = C.c_uint32(0), C.c_uint32(0)
        self._bind_stream()
        L.check(self._lib.pgi_rotation_average_edges(
            self._ctx, _ptr(edges), src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
            None if rows is None else rows.ctypes.data_as(C.c_void_p), P, n_views, C.byref(prm), R.ctypes.data_as(C.c_void_p),
            C.byref(iters), C.byref(used)))
        return R, iters.value, used.value

    # ---- multi-GPU exchange (include/pgi.h: pgi_comm_*, pgi_allgather_edges) ----------------------------------------
    def comm_info(self):
        w, r, k = C.c_uint32(1), C.c_uint32(0), C.c_uint32(0)
        L.check(self._lib.pgi_comm_info(self._ctx, C.byref(w), C.byref(r), C.byref(k)))
        return w.value, r.value, ("none", "rccl", "host")[k.value]

    def allgather_edges(self, local_edges, counts, out=None):
        """local_edges: uint8 [P_r, 200] on this engine's device; counts[r] = records of rank r (counts[rank] == P_r).
        Returns the [sum(counts), 200] table in rank order (asynchronous on the current stream with RCCL)."""
        counts = np.ascontiguousarray(counts, np.uint32)
        total = int(counts.sum())
        if out is None:
            out = torch.empty((total, L.EDGE_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        self._bind_stream()
        L.check(self._lib.pgi_allgather_edges(self._ctx, _ptr(local_edges) if local_edges.numel() else None,
                                              counts.ctypes.data_as(C.c_void_p), _ptr(out)))
        return out

    # ---- descriptor matching (feature_utils.h:135-202) ------------------------------------------
    def prepare_descriptors(self, desc, screen=True):
        """n x 128 float32 (numpy or device tensor) -> prepared image (transposed copy + norms in HBM; with screen=True
        also the padded row-major copy and its half-precision rounding that enable the screened matcher)."""
        d = torch.as_tensor(np.ascontiguousarray(desc, np.float32) if isinstance(desc, np.ndarray) else desc,
                            dtype=torch.float32, device=self.device).contiguous()
        n = int(d.shape[0])
        if n and d.shape[1] != 128:
            raise ValueError("descriptors must be n x 128")
        n_pad = int(self._lib.pgi_desc_padded(n))
        dt = torch.empty((128, max(n_pad, 1)), dtype=torch.float32, device=self.device)
        nrm = torch.empty(max(n_pad, 1), dtype=torch.float32, device=self.device)
        self._bind_stream()
        L.check(self._lib.pgi_desc_prepare(self._ctx, _ptr(d) if n else None, n, _ptr(dt), _ptr(nrm)))
        im = {"t": dt, "norm": nrm, "n": n, "n_pad": n_pad, "rm": None, "f16": None}
        if screen:
            im["rm"] = torch.empty((max(n_pad, 1), 128), dtype=torch.float32, device=self.device)
            im["f16"] = torch.empty((max(n_pad, 1), 128), dtype=torch.int16, device=self.device)
            L.check(self._lib.pgi_desc_prepare_screen(self._ctx, _ptr(d) if n else None, n, _ptr(im["rm"]), _ptr(im["f16"])))
        return im

    def match_descriptors_batch(self, images, pairs, max_matches=None, raw=False):
        """images: list of prepared images; pairs: [(src, dst)] -> per pair (src_idx, dst_idx, ratio) sorted by ratio.
        raw=True returns the device tensors (src, dst, ratio, counts) without a host copy."""
        P = len(pairs)
        if max_matches is None:
            max_matches = max([images[s]["n"] for s, _ in pairs] + [1])
        va, vb = (L.DescView * max(P, 1))(), (L.DescView * max(P, 1))()
        for p, (s, d) in enumerate(pairs):
            for v, im in ((va[p], images[s]), (vb[p], images[d])):
                v.d_desc_t, v.d_norm, v.n, v.n_pad = im["t"].data_ptr(), im["norm"].data_ptr(), im["n"], im["n_pad"]
                v.d_desc_rm = im["rm"].data_ptr() if im.get("rm") is not None else None
                v.d_desc_f16 = im["f16"].data_ptr() if im.get("f16") is not None else None
        src = torch.empty((max(P, 1), max_matches), dtype=torch.int32, device=self.device)
        dst = torch.empty_like(src)
        ratio = torch.empty((max(P, 1), max_matches), dtype=torch.float64, device=self.device)
        counts = torch.zeros(max(P, 1), dtype=torch.int32, device=self.device)
        self._bind_stream()
        L.check(self._lib.pgi_match_descriptors_batch(self._ctx, va, vb, P, max_matches, _ptr(src), _ptr(dst), _ptr(ratio),
                                                      _ptr(counts)))
        if raw:
            return src, dst, ratio, counts
        c = counts.cpu().numpy()
        s_h, d_h, r_h = src.cpu().numpy(), dst.cpu().numpy(), ratio.cpu().numpy()
        return [(s_h[p, :c[p]].astype(np.uint32), d_h[p, :c[p]].astype(np.uint32), r_h[p, :c[p]]) for p in range(P)]

    def upload_keypoints(self, xy, f, w, h):
        """cv::KeyPoint::pt array (n x 2 pixels) + pinhole camera K = [f 0 w/2; 0 f h/2] (pose_graph_builder.h:286)."""
        t = torch.as_tensor(np.ascontiguousarray(xy, np.float32).reshape(-1, 2)).to(self.device)
        return {"xy": t, "n": int(t.shape[0]), "fx": float(f), "fy": float(f), "cx": w / 2.0, "cy": h / 2.0,
                "width": float(w), "height": float(h)}

    def build_correspondences(self, keypoints, pairs, matches, thr_px, top_k=0, dst_uses_src_intrinsics=False, seed=0,
                              pair_id_base=0):
        """createCorrespondenceMatrix on the device (pose_graph_builder.h:864-938): `matches` is the raw output of
        match_descriptors_batch(..., raw=True); returns a batch dict for estimate_pose_batch."""
        src, dst, _, counts = matches
        P, mm = len(pairs), int(src.shape[1])
        va, vb = (L.KeypointView * max(P, 1))(), (L.KeypointView * max(P, 1))()
        for p, (s, d) in enumerate(pairs):
            for v, kp in ((va[p], keypoints[s]), (vb[p], keypoints[d])):
                v.d_xy, v.n = (kp["xy"].data_ptr() if kp["n"] else None), kp["n"]
                v.fx, v.fy, v.cx, v.cy = kp["fx"], kp["fy"], kp["cx"], kp["cy"]
        cap = min(mm, top_k) if top_k else mm
        rows = max(P * cap, 1)
        b = dict(x1=torch.empty(rows, dtype=torch.float32, device=self.device), guesses=None, has_guess=None, n_pairs=P,
                 max_corr=cap, seed=int(seed), pair_id_base=int(pair_id_base))
        for k in ("y1", "x2", "y2"):
            b[k] = torch.empty_like(b["x1"])
        b["offsets"] = torch.zeros(P + 1, dtype=torch.int64, device=self.device)
        b["thr"] = torch.zeros(max(P, 1), dtype=torch.float64, device=self.device)
        self._bind_stream()
        L.check(self._lib.pgi_build_correspondences(
            self._ctx, va, vb, P, mm, _ptr(src), _ptr(dst), _ptr(counts), int(top_k), float(thr_px),
            int(bool(dst_uses_src_intrinsics)), _ptr(b["x1"]), _ptr(b["y1"]), _ptr(b["x2"]), _ptr(b["y2"]), _ptr(b["offsets"]),
            _ptr(b["thr"])))
        return b

    def upload_features(self, xy, desc, f, w, h):
        """Keypoints (n x 2 px) + row-major descriptors (n x 128) of one image, for guided matching."""
        kp = self.upload_keypoints(xy, f, w, h)
        kp["desc"] = torch.as_tensor(np.ascontiguousarray(desc, np.float32).reshape(-1, 128)).to(self.device)
        return kp

    def guided_match_batch(self, features, pairs, poses_Rt, max_n=100, raw=False, n_bins=45):
        """HashingBasedMatcherWithPose::match + the top-N cut of guidedMatching (matcher.h:199-405,
        pose_graph_builder.h:715-783) for every (src, dst) pair with pose (R, t) = poses_Rt[p] (12 doubles).
        n_bins = 45 is the reference's epipolar hashing; n_bins = 0 visits every destination keypoint."""
        P = len(pairs)
        va, vb = (L.FeatureView * max(P, 1))(), (L.FeatureView * max(P, 1))()
        for p, (s, d) in enumerate(pairs):
            for v, ft in ((va[p], features[s]), (vb[p], features[d])):
                v.d_xy = ft["xy"].data_ptr() if ft["n"] else None
                v.d_desc = ft["desc"].data_ptr() if ft["n"] else None
                v.n, v.fx, v.fy, v.cx, v.cy = ft["n"], ft["fx"], ft["fy"], ft["cx"], ft["cy"]
                v.width, v.height = ft.get("width", 2.0 * ft["cx"]), ft.get("height", 2.0 * ft["cy"])
        stride = max_n if max_n else max([features[s]["n"] for s, _ in pairs] + [1])
        pose = np.ascontiguousarray(poses_Rt, np.float64).reshape(max(P, 0), 12)
        src = torch.empty((max(P, 1), stride), dtype=torch.int32, device=self.device)
        dst = torch.empty_like(src)
        ratio = torch.empty((max(P, 1), stride), dtype=torch.float64, device=self.device)
        counts = torch.zeros(max(P, 1), dtype=torch.int32, device=self.device)
        self._bind_stream()
        L.check(self._lib.pgi_guided_match_batch(self._ctx, va, vb, P, pose.ctypes.data_as(C.c_void_p), int(n_bins), int(max_n), stride,
                                                 _ptr(src), _ptr(dst), _ptr(ratio), _ptr(counts)))
        if raw:
            return src, dst, ratio, counts
        c = counts.cpu().numpy()
        s_h, d_h, r_h = src.cpu().numpy(), dst.cpu().numpy(), ratio.cpu().numpy()
        return [(s_h[p, :c[p]].astype(np.uint32), d_h[p, :c[p]].astype(np.uint32), r_h[p, :c[p]]) for p in range(P)]

    # ---- single-pair drop-in (host pointers) --------------------------------------------------
    def estimate_pose(self, corr_aos, thr, guesses=None, seed=0, pair_id=0, min_inliers=0):
        """estimatePose(corr N x 4 f64, thr, guesses) -> (ok, Edge, mask) (pose_graph_builder.h:940-1078).
        min_inliers: the seam's kMinimumInlierNumber_ for this call (0 = the engine's parameter).  Re-entrant:
        concurrent callers get private slots (stream + pinned staging) and overlap on the GPU."""
        c = np.ascontiguousarray(corr_aos, np.float64).reshape(-1, 4)
        g = None if guesses is None else np.ascontiguousarray(guesses, np.float64).reshape(-1, 12)
        e = L.Edge()
        mask = np.zeros(max(len(c), 1), np.uint8)
        self._bind_stream()
        rc = L.check(self._lib.pgi_estimate_pose(
            self._ctx, c.ctypes.data_as(C.c_void_p), len(c), float(thr),
            None if g is None else g.ctypes.data_as(C.c_void_p), 0 if g is None else len(g), int(min_inliers),
            int(seed), int(pair_id), C.byref(e), mask.ctypes.data_as(C.c_void_p)))
        return bool(rc), e, mask[:len(c)]


class DeviceTracklets:
    """reconstruction::Tracklets (point_track.h:541-711) resident in HBM: pgi_tracklets_* of include/pgi.h.
    add() takes one or many (src view, dst view, matches, mask) calls and applies them in order."""

    def __init__(self, engine, n_views):
        self.eng, self._lib = engine, engine._lib
        self._t = self._lib.pgi_tracklets_create(engine._ctx, int(n_views))
        if not self._t:
            raise L.PgiError("pgi_tracklets_create failed: " + L.last_error())
        if not hasattr(engine, "_children"):
            import weakref
            engine._children = weakref.WeakSet()
        engine._children.add(self)  # Engine.close() destroys its stores before its context

    def close(self):
        if getattr(self, "_t", None):
            if getattr(self.eng, "_ctx", None):  # (a store outliving its context is abandoned, never touched)
                self._lib.pgi_tracklets_destroy(self._t)
            self._t = None

    __del__ = close

    def add_batch(self, calls):
        """calls: [(src, dst, matches (m x 2 integer array-like or a pair of device int32 tensors), mask or None)]."""
        dev = self.eng.device
        n = len(calls)
        arr = (L.TrackletPair * max(n, 1))()
        keep = []
        for i, (s, d, m, mask) in enumerate(calls):
            if isinstance(m, tuple) and torch.is_tensor(m[0]):
                ts, td = m
            else:
                mm = np.asarray(m, np.int64).reshape(-1, 2)
                ts = torch.as_tensor(mm[:, 0].astype(np.int32)).to(dev)
                td = torch.as_tensor(mm[:, 1].astype(np.int32)).to(dev)
            tm = None
            if mask is not None:
                tm = mask if torch.is_tensor(mask) else torch.as_tensor(np.asarray(mask, np.uint8)).to(dev)
            keep += [ts, td, tm]
            arr[i].view_src, arr[i].view_dst, arr[i].n_max = int(s), int(d), int(ts.numel())
            arr[i].d_src = ts.data_ptr() if ts.numel() else None
            arr[i].d_dst = td.data_ptr() if td.numel() else None
            arr[i].d_mask = tm.data_ptr() if tm is not None and tm.numel() else None
            arr[i].d_count = None
        self.eng._bind_stream()
        L.check(self._lib.pgi_tracklets_add_batch(self._t, arr, n))

    def add(self, src, dst, matches, mask=None):
        self.add_batch([(src, dst, matches, mask)])

    def get_correspondences_batch(self, queries, max_n, raw=False):
        """[(src view, dst view)] -> per query an (m x 2) array of (source keypoint, destination keypoint)."""
        q = np.asarray(queries, np.uint32).reshape(-1, 2)
        n, stride = len(q), int(max_n) + 1
        vs, vd = np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])
        dev = self.eng.device
        src = torch.empty((max(n, 1), stride), dtype=torch.int32, device=dev)
        dst = torch.empty_like(src)
        cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.eng._bind_stream()
        L.check(self._lib.pgi_tracklets_get_batch(self._t, vs.ctypes.data_as(C.c_void_p), vd.ctypes.data_as(C.c_void_p), n, int(max_n), stride,
                                                  _ptr(src), _ptr(dst), _ptr(cnt)))
        if raw:
            return src, dst, cnt
        c, s_h, d_h = cnt.cpu().numpy(), src.cpu().numpy(), dst.cpu().numpy()
        return [np.stack([s_h[i, :c[i]], d_h[i, :c[i]]], 1).astype(np.int64) for i in range(n)]

    def get_correspondences(self, src, dst, max_n):
        return self.get_correspondences_batch([(src, dst)], max_n)[0]

    def info(self):
        a, b, r = C.c_uint64(), C.c_uint64(), C.c_uint32()
        L.check(self._lib.pgi_tracklets_info(self._t, C.byref(a), C.byref(b), C.byref(r)))
        return {"tracks": a.value, "events": b.value, "rounds": r.value}

    def track(self, index):
        n = C.c_uint32()
        L.check(self._lib.pgi_tracklets_track(self._t, int(index), None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint64)
        L.check(self._lib.pgi_tracklets_track(self._t, int(index), buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return [(int(k >> np.uint64(32)), int(k & np.uint64(0xFFFFFFFF))) for k in buf[:n.value]]
