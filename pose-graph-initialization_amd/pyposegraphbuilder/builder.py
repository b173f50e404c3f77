"""Python mirror of the reference's surface for this path.

* ``PoseGraphBuilder`` -- the 17 constructor arguments of reconstruction::PoseGraphBuilder
  (src/pyposegraphbuilder/include/pose_graph_builder.h:30-47; call site examples/cpp_example.cpp:86-103),
  ``estimatePose`` (the seam, :940-1078) and a wave-scheduled ``run`` over caller-provided candidate pairs.
* ``findEssentialMatrix`` -- the only historical hint of the Python API
  (src/pyposegraphbuilder/src/bindings.cpp:74-162, 255-266, commented out upstream).
"""
import numpy as np

from .engine import Engine


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

    def run(self, pairs, waveSize=4096, seed=0):
        """pairs: iterable of dict(src, dst, similarity, correspondences[N,4], threshold).
        Returns the pose graph {(src, dst): dict(R, t, score)}; score = inliers / matches (:645-654)."""
        cand = [p for p in pairs if p["similarity"] >= self.kSimilarityThreshold
                and len(p["correspondences"]) >= self.kMinimumPointNumber]          # :550-551
        cand.sort(key=lambda p: (-p["similarity"], p["src"], p["dst"]))             # heap order
        graph = {}
        for w0 in range(0, len(cand), waveSize):
            wave = [p for p in cand[w0:w0 + waveSize] if (p["src"], p["dst"]) not in graph]   # :426-431
            if not wave:
                continue
            c = [np.asarray(p["correspondences"], np.float32) for p in wave]
            off = np.concatenate([[0], np.cumsum([len(x) for x in c])]).astype(np.uint64)
            allc = np.concatenate(c)
            b = self.engine.upload(allc[:, 0], allc[:, 1], allc[:, 2], allc[:, 3], off,
                                   np.array([p["threshold"] for p in wave]), seed=seed + w0)
            edges, _ = self.engine.estimate_pose_batch(b)
            for p, e, n in zip(wave, self.engine.edges_to_numpy(edges), np.diff(off.astype(np.int64))):
                if e["status"] == 1:
                    graph[(p["src"], p["dst"])] = dict(R=e["R"].reshape(3, 3).copy(), t=e["t"].copy(),
                                                       score=float(e["n_inl"]) / float(max(n, 1)))
        return graph


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
