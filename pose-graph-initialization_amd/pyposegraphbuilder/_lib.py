"""ctypes binding of libpgi.so (include/pgi.h).  No CPU fallback: importing the engine
without the built HIP library, or creating a context without a GPU, raises."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG), "libpgi.so")
DBG_DOUBLES = 358


class Params(C.Structure):
    _fields_ = [("confidence", C.c_double), ("max_iters", C.c_uint32), ("round_size", C.c_uint32),
                ("lo_iters", C.c_uint32), ("min_inliers", C.c_uint32), ("fixed_budget", C.c_uint32),
                ("guess_quirk", C.c_uint32), ("vote_all_rows", C.c_uint32), ("guess_mode", C.c_uint32),
                ("lo_linear_pct", C.c_uint32), ("sampler", C.c_uint32), ("lo_graph_cut", C.c_uint32)]


class Edge(C.Structure):
    _fields_ = [("E", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("status", C.c_int32), ("n_inl", C.c_uint32), ("score", C.c_uint32),
                ("iters", C.c_uint32), ("votes", C.c_uint32), ("cand", C.c_uint32),
                ("used_guess", C.c_uint32), ("lo_runs", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("d_x1", C.c_void_p), ("d_y1", C.c_void_p), ("d_x2", C.c_void_p), ("d_y2", C.c_void_p),
                ("d_offsets", C.c_void_p), ("d_thr", C.c_void_p), ("d_guess_Rt", C.c_void_p),
                ("d_has_guess", C.c_void_p), ("n_pairs", C.c_uint32), ("max_corr", C.c_uint32),
                ("pair_id_base", C.c_uint64), ("seed", C.c_uint64)]


class RotEdge(C.Structure):
    _fields_ = [("src", C.c_uint32), ("dst", C.c_uint32), ("R", C.c_double * 9), ("weight", C.c_double)]


class RotAvgParams(C.Structure):
    _fields_ = [("l1_iters", C.c_uint32), ("irls_iters", C.c_uint32), ("cg_iters", C.c_uint32),
                ("sigma_deg", C.c_double), ("tol", C.c_double)]


EDGE_DTYPE = np.dtype([("E", "f8", 9), ("R", "f8", 9), ("t", "f8", 3), ("status", "i4"),
                       ("n_inl", "u4"), ("score", "u4"), ("iters", "u4"), ("votes", "u4"),
                       ("cand", "u4"), ("used_guess", "u4"), ("lo_runs", "u4")])
ROT_EDGE_DTYPE = np.dtype([("src", "u4"), ("dst", "u4"), ("R", "f8", 9), ("weight", "f8")])
assert EDGE_DTYPE.itemsize == C.sizeof(Edge) == 200
assert ROT_EDGE_DTYPE.itemsize == C.sizeof(RotEdge)

# every symbol include/pgi.h declares
SYMBOLS = ["pgi_last_error", "pgi_device_count", "pgi_default_params", "pgi_create", "pgi_destroy",
           "pgi_set_stream", "pgi_get_stream", "pgi_get_device", "pgi_set_params", "pgi_synchronize", "pgi_estimate_pose_batch", "pgi_estimate_pose_batch_host",
           "pgi_estimate_pose", "pgi_score_pose_batch", "pgi_score_pose_f64", "pgi_score_pose_f64_host", "pgi_decompose_batch", "pgi_pose_from_essential_host", "pgi_screen_guesses",
           "pgi_five_point_batch", "pgi_default_rotavg_params", "pgi_rotation_average", "pgi_desc_padded",
           "pgi_desc_prepare", "pgi_desc_prepare_screen", "pgi_match_descriptors_batch", "pgi_build_correspondences", "pgi_guided_match_batch",
           "pgi_get_params", "pgi_rotation_average_edges", "pgi_comm_unique_id", "pgi_comm_init_rccl", "pgi_comm_init_host",
           "pgi_comm_destroy", "pgi_comm_info", "pgi_comm_rccl_probe", "pgi_allgather_edges", "pgi_allgatherv", "pgi_host_register", "pgi_host_unregister",
           "pgi_tracklets_create", "pgi_tracklets_destroy", "pgi_tracklets_add_batch", "pgi_tracklets_get_batch",
           "pgi_tracklets_info", "pgi_tracklets_track"]
COMM_ID_BYTES = 128
# pgi_allgatherv_fn: int (*)(void* user, const void* send, uint64 send_bytes, void* recv, const uint64* recv_bytes, uint32 world)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32)

_lib = None


class DescView(C.Structure):
    """pgi_desc_view (include/pgi.h)."""
    _fields_ = [("d_desc_t", C.c_void_p), ("d_norm", C.c_void_p), ("n", C.c_uint32), ("n_pad", C.c_uint32),
                ("d_desc_rm", C.c_void_p), ("d_desc_f16", C.c_void_p)]


class KeypointView(C.Structure):
    """pgi_keypoint_view (include/pgi.h)."""
    _fields_ = [("d_xy", C.c_void_p), ("n", C.c_uint32), ("reserved", C.c_uint32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


class TrackletPair(C.Structure):
    """pgi_tracklet_pair (include/pgi.h)."""
    _fields_ = [("view_src", C.c_uint32), ("view_dst", C.c_uint32), ("n_max", C.c_uint32), ("reserved", C.c_uint32),
                ("d_src", C.c_void_p), ("d_dst", C.c_void_p), ("d_mask", C.c_void_p), ("d_count", C.c_void_p)]


class FeatureView(C.Structure):
    """pgi_feature_view (include/pgi.h)."""
    _fields_ = [("d_xy", C.c_void_p), ("d_desc", C.c_void_p), ("n", C.c_uint32), ("reserved", C.c_uint32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("width", C.c_double), ("height", C.c_double)]


class PgiError(RuntimeError):
    pass


def load():
    """Loads libpgi.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgiError("libpgi.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C pose-graph-initialization_amd`; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.pgi_last_error.restype = C.c_char_p
    lib.pgi_create.restype = C.c_void_p
    lib.pgi_create.argtypes = [C.c_int, C.POINTER(Params)]
    lib.pgi_destroy.argtypes = [C.c_void_p]
    lib.pgi_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.pgi_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.pgi_get_device.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.pgi_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    lib.pgi_synchronize.argtypes = [C.c_void_p]
    lib.pgi_desc_padded.restype = C.c_uint32
    lib.pgi_desc_padded.argtypes = [C.c_uint32]
    lib.pgi_desc_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.pgi_desc_prepare_screen.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.pgi_match_descriptors_batch.argtypes = [C.c_void_p, C.POINTER(DescView), C.POINTER(DescView), C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_build_correspondences.argtypes = [C.c_void_p, C.POINTER(KeypointView), C.POINTER(KeypointView), C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_guided_match_batch.argtypes = [C.c_void_p, C.POINTER(FeatureView), C.POINTER(FeatureView), C.c_uint32, C.c_void_p,
                                           C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_estimate_pose_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]
    lib.pgi_estimate_pose_batch_host.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.pgi_estimate_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_uint64, C.c_uint64, C.POINTER(Edge), C.c_void_p]
    lib.pgi_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    lib.pgi_tracklets_create.restype = C.c_void_p
    lib.pgi_tracklets_create.argtypes = [C.c_void_p, C.c_uint32]
    lib.pgi_tracklets_destroy.restype = None
    lib.pgi_tracklets_destroy.argtypes = [C.c_void_p]
    lib.pgi_tracklets_add_batch.argtypes = [C.c_void_p, C.POINTER(TrackletPair), C.c_uint32]
    lib.pgi_tracklets_get_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    lib.pgi_tracklets_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.pgi_tracklets_track.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.pgi_host_register.argtypes = [C.c_void_p, C.c_uint64]
    lib.pgi_host_unregister.argtypes = [C.c_void_p]
    lib.pgi_rotation_average_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.POINTER(RotAvgParams), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_comm_unique_id.argtypes = [C.c_void_p]
    lib.pgi_comm_init_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pgi_comm_init_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, ALLGATHERV_FN, C.c_void_p]
    lib.pgi_comm_destroy.argtypes = [C.c_void_p]
    lib.pgi_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pgi_allgather_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_allgatherv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_score_pose_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
    lib.pgi_score_pose_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    lib.pgi_score_pose_f64_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double, C.c_uint32,
                                            C.POINTER(C.c_uint32), C.c_void_p]
    lib.pgi_comm_rccl_probe.argtypes = [C.POINTER(C.c_int)]
    lib.pgi_pose_from_essential_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pgi_screen_guesses.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.pgi_decompose_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_five_point_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgi_rotation_average.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.POINTER(RotAvgParams), C.c_void_p, C.c_void_p]
    lib.pgi_default_rotavg_params.argtypes = [C.POINTER(RotAvgParams)]
    _lib = lib
    return lib


def last_error():
    return load().pgi_last_error().decode()


def check(rc):
    if rc < 0:
        raise PgiError("pgi error %d: %s" % (rc, last_error()))
    return rc


def default_params(**kw):
    p = Params()
    load().pgi_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


def kernel_source_sha256(files=("csrc/pgi_kernels.hip", "csrc/pgi_device.hpp")):
    """sha256 over the sources of the pose-estimation kernels (K1/K2/K3), in the given order.  Profile summaries under
    profiles/ record it (scripts/k1_pmc_json.py); bench.py replays their counters only while it still matches the
    sources the loaded library was built from."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_PKG)
    for f in files:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
