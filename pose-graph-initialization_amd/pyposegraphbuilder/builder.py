"""Python mirror of the reference's surface for this path.

* ``PoseGraphBuilder`` -- the 17 constructor arguments of reconstruction::PoseGraphBuilder
  (src/pyposegraphbuilder/include/pose_graph_builder.h:30-47; call site examples/cpp_example.cpp:86-103),
  ``estimatePose`` (the seam, :940-1078) and a wave-scheduled ``run`` over caller-provided candidate pairs.
* ``findEssentialMatrix`` -- the only historical hint of the Python API
  (src/pyposegraphbuilder/src/bindings.cpp:74-162, 255-266, commented out upstream).
"""
import numpy as np

import ctypes as _C

from .engine import Engine


class _HostConfig(_C.Structure):
    """pgih_config (include/pgi_host.h): the 17 constructor arguments in the reference's order"""
    _fields_ = [(n, _C.c_uint64) for n in ("core_number", "maximum_tracklet_number", "maximum_search_depth", "maximum_path_number",
                                           "minimum_inlier_number", "minimum_point_number", "maximum_point_number_for_epipolar_hashing")] + \
               [(n, _C.c_double) for n in ("traversal_heuristics_weight", "similarity_threshold", "inlier_outlier_threshold")] + \
               [(n, _C.c_char_p) for n in ("image_path", "workspace_path", "similarity_graph_path", "focal_length_path")] + \
               [(n, _C.c_int32) for n in ("use_path_finding", "use_gpu", "use_epipolar_hashing")]


class _HostView(_C.Structure):  # pgih_view
    _fields_ = [("keypoints", _C.c_void_p), ("descriptors", _C.c_void_p), ("n", _C.c_uint32), ("pad", _C.c_uint32),
                ("focal_length", _C.c_double), ("width", _C.c_double), ("height", _C.c_double)]


_GRAPH_EDGE = np.dtype([("src", "<u4"), ("dst", "<u4"), ("score", "<f8"), ("R", "<f8", 9), ("t", "<f8", 3)])  # pgih_graph_edge


def bindProcessToDeviceNode(device=-1):
    """One process per GPU: keep this process -- the calling thread and every thread it starts later -- on the CPUs of the NUMA
    node the HIP device hangs on (device < 0: the current one), what `numactl --cpunodebind` does in a launcher; arrays created
    afterwards land on that node.  Call it before the inputs are created.  Returns the node, or -1 when nothing was done
    (pgih_bind_process_to_device_node of libpgi_host.so)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libpgi_host.so")
    if not os.path.exists(path):
        raise RuntimeError("libpgi_host.so not built (%s): make -C pose-graph-initialization_amd" % path)
    lib = C.CDLL(path)
    lib.pgih_bind_process_to_device_node.argtypes = [C.c_int]
    return int(lib.pgih_bind_process_to_device_node(int(device)))


class PoseGraphBuilder:
    def __init__(self, kCoreNumber=20, kMaximumTrackletNumber=5000, kMaximumSearchDepth=5, kMaximumPathNumber=100,
                 kMinimumInlierNumber=20, kMinimumPointNumber=50, kMaximumPointNumberForEpipolarHashing=100,
                 kTraversalHeuristicsWeight=0.8, kSimilarityThreshold=0.5, kInlierOutlierThreshold=0.4,
                 kImagePath="", kWorkspacePath="", kSimilarityGraphPath="", kFocalLengthPath="",
                 kUsePathFinding=True, kUseGPU=True, kUseEpipolarHashing=True):
        # defaults: examples/cpp_example.cpp:32-66
        self.kCoreNumber = kCoreNumber
        self.kMaximumTrackletNumber = kMaximumTrackletNumber
        self.kMaximumSearchDepth = kMaximumSearchDepth
        self.kMaximumPathNumber = kMaximumPathNumber
        self.kMinimumInlierNumber = kMinimumInlierNumber
        self.kMinimumPointNumber = kMinimumPointNumber
        self.kMaximumPointNumberForEpipolarHashing = kMaximumPointNumberForEpipolarHashing
        self.kTraversalHeuristicsWeight = kTraversalHeuristicsWeight
        self.kSimilarityThreshold = kSimilarityThreshold
        self.kInlierOutlierThreshold = kInlierOutlierThreshold
        self.kImagePath, self.kWorkspacePath = kImagePath, kWorkspacePath
        self.kSimilarityGraphPath, self.kFocalLengthPath = kSimilarityGraphPath, kFocalLengthPath
        self.kUsePathFinding, self.kUseGPU, self.kUseEpipolarHashing = kUsePathFinding, kUseGPU, kUseEpipolarHashing
        self.engine = Engine(min_inliers=kMinimumInlierNumber)

    def estimatePose(self, correspondences, threshold, poseGuesses=None, seed=0, pairId=0):
        """(ok, R[3,3], t[3], inlierMask uint8[N], inlierNumber) -- pose_graph_builder.h:940-1078."""
        ok, e, mask = self.engine.estimate_pose(correspondences, threshold, poseGuesses, seed=seed, pair_id=pairId)
        return ok, np.array(e.R).reshape(3, 3), np.array(e.t), mask, int(e.n_inl)

    def run(self, pairs, waveSize=4096, seed=0, *, rotationGuided=False, numViews=0, graphCut=0):
        """pairs: iterable of dict(src, dst, similarity, correspondences[N,4], threshold).
        Returns the pose graph {(src, dst): dict(R, t, score)}; score = inliers / matches (:645-654).
        seed: wave w of the run draws its hypotheses with seed + w.  rotationGuided (keyword only): BASELINE config 5's
        re-estimation of chained poses, set on the C++ builder for THIS call and restored afterwards.  numViews: every
        view id must be below it (0 = derived from the ids).  graphCut (keyword only): lambda * 64 of the graph-cut local
        optimisation (pgi_params.lo_graph_cut; 9 ~ 0.14; 0 = off), for this call.

        This IS the C++ scheduler (host/pose_graph_builder.cpp PoseGraphBuilder::run behind pgih_run_pairs of
        libpgi_host.so, include/pgi_host.h): descending-similarity waves, and -- with kUsePathFinding -- A* pose guesses
        on the graph committed by the previous waves, visibility table, batched guess screening and estimatePose.  The
        Python class only marshals arrays; `statistics` holds the run's counters afterwards."""
        import ctypes as C
        pairs = list(pairs)
        P = len(pairs)
        src = np.array([p["src"] for p in pairs], np.uint32)
        dst = np.array([p["dst"] for p in pairs], np.uint32)
        sim = np.array([p["similarity"] for p in pairs], np.float64)
        thr = np.array([p["threshold"] for p in pairs], np.float64)
        rows = [np.ascontiguousarray(p["correspondences"], np.float64).reshape(-1, 4) for p in pairs]
        off = np.zeros(P + 1, np.uint64)
        off[1:] = np.cumsum([len(r) for r in rows])
        corr = np.concatenate(rows) if P else np.zeros((0, 4))
        lib, h = self._host()
        if lib.pgih_set_rotation_guided(h, int(bool(rotationGuided))) < 0 or lib.pgih_set_graph_cut(h, int(graphCut)) < 0:
            raise RuntimeError(lib.pgih_last_error().decode())
        edges = np.zeros(max(P, 1), _GRAPH_EDGE)
        n_edges = C.c_uint32(0)
        stats = np.zeros(16, np.uint64)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        try:
            rc = lib.pgih_run_pairs(h, int(numViews), P, ptr(src), ptr(dst), ptr(sim), ptr(thr), ptr(off), ptr(corr), int(waveSize),
                                    int(seed), ptr(edges), len(edges), C.byref(n_edges), ptr(stats))
            err = lib.pgih_last_error().decode() if rc < 0 else None
        finally:
            if rotationGuided:  # the switches belong to this call, not to the builder
                lib.pgih_set_rotation_guided(h, 0)
            if graphCut:
                lib.pgih_set_graph_cut(h, 0)
        if rc < 0:
            raise RuntimeError(err)
        self.statistics = dict(zip(("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes",
                                    "poses_from_guess", "hypotheses", "waves", "graph_edges", "quirk_only_guesses"),
                                   (int(v) for v in stats)))
        return {(int(e["src"]), int(e["dst"])): dict(R=e["R"].reshape(3, 3).copy(), t=e["t"].copy(), score=float(e["score"]))
                for e in edges[:n_edges.value]}

    def runFeatures(self, views, pairs, waveSize=512, *, rotationGuided=False, deviceTracklets=True):
        """The loop body of processImages (pose_graph_builder.h:391-709) on in-memory features -- the C++
        PoseGraphBuilder::processFeatures behind pgih_run_features (include/pgi_host.h), marshalled only.

        views: sequence of dict(xy[n,2] float32 pixels, desc[n,128] float32, focal, width, height);
        pairs: iterable of (src, dst, similarity).  Returns the pose graph {(src, dst): dict(R, t, score)};
        `statistics` holds the counters, `stage_seconds` the wall-clock split of the run."""
        import ctypes as C
        keep = []   # the arrays the view records point into
        vrec = (_HostView * max(len(views), 1))()
        for k, v in enumerate(views):
            xy = np.ascontiguousarray(v["xy"], np.float32).reshape(-1, 2)
            desc = np.ascontiguousarray(v["desc"], np.float32).reshape(-1, 128)
            if len(xy) != len(desc):
                raise ValueError("view %d: %d keypoints but %d descriptors" % (k, len(xy), len(desc)))
            keep += [xy, desc]
            vrec[k] = _HostView(xy.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), len(xy), 0, float(v["focal"]),
                                float(v["width"]), float(v["height"]))
        pairs = list(pairs)
        P = len(pairs)
        src = np.array([p[0] for p in pairs], np.uint32)
        dst = np.array([p[1] for p in pairs], np.uint32)
        sim = np.array([p[2] for p in pairs], np.float64)
        lib, h = self._host()
        if lib.pgih_set_rotation_guided(h, int(bool(rotationGuided))) < 0:
            raise RuntimeError(lib.pgih_last_error().decode())
        edges = np.zeros(max(P, 1), _GRAPH_EDGE)
        n_edges = C.c_uint32(0)
        stats = np.zeros(24, np.uint64)
        stages = np.zeros(8, np.float64)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        try:
            rc = lib.pgih_run_features(h, len(views), vrec, P, ptr(src), ptr(dst), ptr(sim), int(waveSize), int(bool(deviceTracklets)), ptr(edges),
                                       len(edges), C.byref(n_edges), ptr(stats), ptr(stages))
            err = lib.pgih_last_error().decode() if rc < 0 else None
        finally:
            if rotationGuided:  # the switch belongs to this call, not to the builder
                lib.pgih_set_rotation_guided(h, 0)
        if rc < 0:
            raise RuntimeError(err)
        names = ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses",
                 "waves", "graph_edges", "quirk_only_guesses")
        self.statistics = dict(zip(names, (int(v) for v in stats)))
        self.statistics.update(zip(("matching_runs", "quick_matching_runs", "guided_matching_runs", "guided_matches_added", "track_number",
                                    "too_few_matches"), (int(v) for v in stats[16:22])))
        self.stage_seconds = dict(zip(("upload + prepare", "quick matching", "matching", "correspondences", "A*", "pose estimation",
                                       "guided", "commit + tracklets"), (float(v) for v in stages)))
        return {(int(e["src"]), int(e["dst"])): dict(R=e["R"].reshape(3, 3).copy(), t=e["t"].copy(), score=float(e["score"]))
                for e in edges[:n_edges.value]}

    def _host(self):
        """libpgi_host.so's builder with this object's 17 constructor arguments (created on first use)."""
        import ctypes as C
        import os
        if getattr(self, "_h", None):
            return self._hlib, self._h
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libpgi_host.so")
        if not os.path.exists(path):
            raise RuntimeError("libpgi_host.so not built (%s): make -C pose-graph-initialization_amd" % path)
        lib = C.CDLL(path)
        lib.pgih_last_error.restype = C.c_char_p
        lib.pgih_create.restype = C.c_void_p
        lib.pgih_create.argtypes = [C.POINTER(_HostConfig)]
        lib.pgih_destroy.argtypes = [C.c_void_p]
        lib.pgih_set_rotation_guided.argtypes = [C.c_void_p, C.c_int]
        lib.pgih_set_graph_cut.argtypes = [C.c_void_p, C.c_uint32]
        lib.pgih_set_progressive_sampling.argtypes = [C.c_void_p, C.c_int]
        lib.pgih_run_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32,
                                                                                                 C.POINTER(C.c_uint32), C.c_void_p]
        lib.pgih_run_features.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        enc = lambda s_: str(s_).encode()
        cfg = _HostConfig(self.kCoreNumber, self.kMaximumTrackletNumber, self.kMaximumSearchDepth, self.kMaximumPathNumber,
                          self.kMinimumInlierNumber, self.kMinimumPointNumber, self.kMaximumPointNumberForEpipolarHashing,
                          self.kTraversalHeuristicsWeight, self.kSimilarityThreshold, self.kInlierOutlierThreshold,
                          enc(self.kImagePath), enc(self.kWorkspacePath), enc(self.kSimilarityGraphPath), enc(self.kFocalLengthPath),
                          int(self.kUsePathFinding), int(self.kUseGPU), int(self.kUseEpipolarHashing))
        h = lib.pgih_create(C.byref(cfg))
        if not h:
            raise RuntimeError(lib.pgih_last_error().decode())
        self._hlib, self._h = lib, h
        return lib, h

    def close(self):
        if getattr(self, "_h", None):
            self._hlib.pgih_destroy(self._h)
            self._h = None
        if getattr(self, "engine", None) is not None:
            self.engine.close()

    __del__ = close


_default_engine = None


def findEssentialMatrix(x1y1, x2y2, K1, K2, h1=0, w1=0, h2=0, w2=0, threshold=1.0, conf=0.99, max_iters=10000):
    """(E[3,3] or None, mask bool[n]); threshold in pixels, points in pixels, like the historical binding."""
    global _default_engine
    x1y1, x2y2 = np.asarray(x1y1, np.float64), np.asarray(x2y2, np.float64)
    K1, K2 = np.asarray(K1, np.float64), np.asarray(K2, np.float64)
    if x1y1.ndim != 2 or x1y1.shape[1] != 2 or x2y2.shape != x1y1.shape:
        raise ValueError("x1y1 and x2y2 must both be [n,2]")
    if K1.shape != (3, 3) or K2.shape != (3, 3):
        raise ValueError("K1 and K2 must be [3,3]")
    if _default_engine is None:
        _default_engine = Engine()
    _default_engine.set_params(confidence=conf, max_iters=int(max_iters))
    n1 = (x1y1 - K1[:2, 2]) / np.array([K1[0, 0], K1[1, 1]])
    n2 = (x2y2 - K2[:2, 2]) / np.array([K2[0, 0], K2[1, 1]])
    thr = threshold / ((K1[0, 0] + K1[1, 1] + K2[0, 0] + K2[1, 1]) / 4.0)   # pose_graph_builder.h:934-937
    ok, e, mask = _default_engine.estimate_pose(np.concatenate([n1, n2], 1), thr)
    if not ok:
        return None, mask.astype(bool)
    return np.array(e.E).reshape(3, 3), mask.astype(bool)
