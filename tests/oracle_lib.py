"""ctypes binding of the CPU oracle (oracle/libpgi_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
MAX_MODELS = 10


class Params(C.Structure):
    _fields_ = [("confidence", C.c_double), ("max_iters", C.c_uint32),
                ("round_size", C.c_uint32), ("lo_iters", C.c_uint32),
                ("min_inliers", C.c_uint32), ("fixed_budget", C.c_uint32),
                ("guess_quirk", C.c_uint32), ("vote_all_rows", C.c_uint32),
                ("guess_mode", C.c_uint32), ("lo_linear_pct", C.c_uint32), ("sampler", C.c_uint32), ("lo_graph_cut", C.c_uint32)]


class Edge(C.Structure):
    _fields_ = [("E", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("status", C.c_int32), ("n_inl", C.c_uint32), ("score", C.c_uint32),
                ("iters", C.c_uint32), ("votes", C.c_uint32), ("cand", C.c_uint32),
                ("used_guess", C.c_uint32), ("lo_runs", C.c_uint32)]


class BackendDbg(C.Structure):
    _fields_ = [("cons", C.c_double * 200), ("red", C.c_double * 100),
                ("poly", C.c_double * 11), ("roots", C.c_double * MAX_MODELS),
                ("n_roots", C.c_uint32)]


EDGE_DTYPE = np.dtype([("E", "f8", 9), ("R", "f8", 9), ("t", "f8", 3), ("status", "i4"),
                       ("n_inl", "u4"), ("score", "u4"), ("iters", "u4"), ("votes", "u4"),
                       ("cand", "u4"), ("used_guess", "u4"), ("lo_runs", "u4")])
assert EDGE_DTYPE.itemsize == C.sizeof(Edge)

_lib = None


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libpgi_oracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("pgi_oracle.c", "pgi_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.pgo_ref_sampson_sq.restype = C.c_double
        _lib.pgo_mix64.restype = C.c_uint64
        _lib.pgo_mix64.argtypes = [C.c_uint64]
        _lib.pgo_num_threads.restype = C.c_int
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_params(**kw):
    p = Params()
    lib().pgo_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ref_sampson_sq(corr4, E):
    c, E = f64(corr4), f64(E)
    return lib().pgo_ref_sampson_sq(_p(c), _p(E))


def ref_get_inliers(corr, E, thr):
    corr, E = f64(corr), f64(E)
    idx = np.zeros(len(corr), np.uint32)
    k = lib().pgo_ref_get_inliers(_p(corr), C.c_uint32(len(corr)), _p(E), C.c_double(thr), _p(idx))
    return idx[:k]


def ref_pose_test(corr, R, t, thr, min_inl):
    corr, R, t = f64(corr), f64(R), f64(t)
    n = C.c_uint32(0)
    ok = lib().pgo_ref_pose_test(_p(corr), C.c_uint32(len(corr)), _p(R), _p(t), C.c_double(thr),
                                 C.c_uint32(min_inl), C.byref(n))
    return bool(ok), n.value


def ref_essential_from_pose(R, t):
    R, t = f64(R), f64(t)
    E = np.zeros(9)
    lib().pgo_ref_essential_from_pose(_p(R), _p(t), _p(E))
    return E.reshape(3, 3)


def ref_chain_pose(Re, te, inverted, R, t):
    R, t = f64(R).copy(), f64(t).copy()
    lib().pgo_ref_chain_pose(_p(f64(Re)), _p(f64(te)), C.c_int(int(inverted)), _p(R), _p(t))
    return R.reshape(3, 3), t


def ref_normalize_corr(ks, kd, ms, md, cam_src, cam_dst, src_for_dst, thr_px):
    ks, kd = f32(ks), f32(kd)
    ms, md = np.ascontiguousarray(ms, np.uint32), np.ascontiguousarray(md, np.uint32)
    out = np.zeros((len(ms), 4))
    thr = C.c_double(0)
    lib().pgo_ref_normalize_corr(_p(ks), _p(kd), _p(ms), _p(md), C.c_uint32(len(ms)),
                                 *[C.c_double(v) for v in cam_src], *[C.c_double(v) for v in cam_dst],
                                 C.c_int(int(src_for_dst)), C.c_double(thr_px), _p(out), C.byref(thr))
    return out, thr.value


def sample5(seed, pair_id, hyp, n):
    idx = np.zeros(5, np.uint32)
    lib().pgo_sample5(C.c_uint64(seed), C.c_uint64(pair_id), C.c_uint32(hyp), C.c_uint32(n), _p(idx))
    return idx


def score_model(E, x1, y1, x2, y2, thr):
    E = f32(E).ravel()
    s, c = C.c_uint32(0), C.c_uint32(0)
    lib().pgo_score_model(_p(E), _p(x1), _p(y1), _p(x2), _p(y2), C.c_uint32(len(x1)),
                          C.c_double(thr), C.byref(s), C.byref(c))
    return s.value, c.value


def mask_model(E, x1, y1, x2, y2, tau2):
    E = f32(E).ravel()
    m = np.zeros(len(x1), np.uint8)
    c = lib().pgo_mask_model(_p(E), _p(x1), _p(y1), _p(x2), _p(y2), C.c_uint32(len(x1)),
                             C.c_float(tau2), _p(m))
    return m, c


def nullspace5(pts):
    pts = f32(pts)
    b = np.zeros(36)
    lib().pgo_nullspace5(_p(pts), _p(b))
    return b.reshape(4, 9)


def backend(basis, sample=None):
    basis = f64(basis)
    models = np.zeros((MAX_MODELS, 9), np.float32)
    dbg = BackendDbg()
    s = f32(sample) if sample is not None else None
    n = lib().pgo_backend(_p(basis), _p(s), C.c_uint32(0 if s is None else len(s)), _p(models),
                          C.byref(dbg))
    return models[:n], dbg


def five_point(pts):
    pts = f32(pts)
    models = np.zeros((MAX_MODELS, 9), np.float32)
    dbg = BackendDbg()
    n = lib().pgo_five_point(_p(pts), _p(models), C.byref(dbg))
    return models[:n], dbg


def normal_matrix(x1, y1, x2, y2, mask=None):
    A = np.zeros(81)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    lib().pgo_normal_matrix(_p(x1), _p(y1), _p(x2), _p(y2), _p(m), C.c_uint32(len(x1)), _p(A))
    return A.reshape(9, 9)


def jacobi9(A):
    A = f64(A).copy().ravel()
    V = np.zeros(81)
    lib().pgo_jacobi9(_p(A), _p(V))
    return A.reshape(9, 9), V.reshape(9, 9)


def basis_from_eigen(A, V):
    b = np.zeros(36)
    lib().pgo_basis_from_eigen(_p(f64(A)), _p(f64(V)), _p(b))
    return b.reshape(4, 9)


def npoint(x1, y1, x2, y2, mask=None):
    models = np.zeros((MAX_MODELS, 9), np.float32)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    n = lib().pgo_npoint(_p(x1), _p(y1), _p(x2), _p(y2), _p(m), C.c_uint32(len(x1)), _p(models))
    return models[:n]


def svd3(E):
    E = f64(E).ravel()
    U, S, V = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().pgo_svd3(_p(E), _p(U), _p(S), _p(V))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def decompose(E, x1, y1, x2, y2, mask, vote_all=False):
    E = f64(E).ravel()
    R, t = np.zeros(9), np.zeros(3)
    votes = np.zeros(4, np.uint32)
    cand = C.c_uint32(0)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    lib().pgo_decompose(_p(E), _p(x1), _p(y1), _p(x2), _p(y2), _p(m), C.c_uint32(len(x1)),
                        C.c_int(int(vote_all)), _p(R), _p(t), _p(votes), C.byref(cand))
    return R.reshape(3, 3), t, votes, cand.value


def ransac_essential(x1, y1, x2, y2, thr, prm, seed, pair_id):
    out = Edge()
    mask = np.zeros(len(x1), np.uint8)
    lib().pgo_ransac_essential(_p(x1), _p(y1), _p(x2), _p(y2), C.c_uint32(len(x1)),
                               C.c_double(thr), C.byref(prm), C.c_uint64(seed),
                               C.c_uint64(pair_id), C.byref(out), _p(mask))
    return out, mask


def estimate_pose(x1, y1, x2, y2, thr, guess, prm, seed, pair_id):
    out = Edge()
    mask = np.zeros(len(x1), np.uint8)
    g = f64(guess).ravel() if guess is not None else None
    lib().pgo_estimate_pose(_p(x1), _p(y1), _p(x2), _p(y2), C.c_uint32(len(x1)), C.c_double(thr),
                            _p(g), C.byref(prm), C.c_uint64(seed), C.c_uint64(pair_id),
                            C.byref(out), _p(mask))
    return out, mask


def estimate_pose_batch(x1, y1, x2, y2, offsets, thr, prm, seed, pair_id_base=0, guesses=None,
                        has_guess=None, threads=0):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    P = len(offsets) - 1
    thr = f64(np.broadcast_to(thr, (P,)))
    out = np.zeros(P, EDGE_DTYPE)
    masks = np.zeros(len(x1), np.uint8)
    g = f64(guesses) if guesses is not None else None
    hg = np.ascontiguousarray(has_guess, np.uint8) if has_guess is not None else None
    lib().pgo_estimate_pose_batch(_p(x1), _p(y1), _p(x2), _p(y2), _p(offsets), C.c_uint32(P),
                                  _p(thr), _p(g), _p(hg), C.byref(prm), C.c_uint64(seed),
                                  C.c_uint64(pair_id_base), _p(out), _p(masks), C.c_int(threads))
    return out, masks


def match_descriptors(A, B):
    A, B = f32(A), f32(B)
    k1, k2, d = len(A), len(B), A.shape[1]
    oi, oj, orr = np.zeros(k1, np.uint32), np.zeros(k1, np.uint32), np.zeros(k1)
    lib().pgo_match_descriptors.restype = C.c_uint32
    m = lib().pgo_match_descriptors(_p(A), C.c_uint32(k1), _p(B), C.c_uint32(k2), C.c_uint32(d), _p(oi), _p(oj), _p(orr))
    return oi[:m], oj[:m], orr[:m]


def fundamental_from_essential(E, k_src, k_dst):
    F = np.zeros(9)
    lib().pgo_fundamental_from_essential(_p(f64(E).ravel()), _p(f64(k_src)), _p(f64(k_dst)), _p(F))
    return F


def guided_match(F, kp1, kp2, d1, d2):
    kp1, kp2, d1, d2 = f32(kp1), f32(kp2), f32(d1), f32(d2)
    n1, n2 = len(kp1), len(kp2)
    oi, oj, orr = np.zeros(max(n1, 1), np.uint32), np.zeros(max(n1, 1), np.uint32), np.zeros(max(n1, 1))
    lib().pgo_guided_match.restype = C.c_uint32
    m = lib().pgo_guided_match(_p(f64(F).ravel()), _p(kp1), C.c_uint32(n1), _p(kp2), C.c_uint32(n2), _p(d1), _p(d2),
                               C.c_uint32(d1.shape[1] if n1 else 128), _p(oi), _p(oj), _p(orr))
    return oi[:m], oj[:m], orr[:m]


def ref_pose_from_essential(E, corr_aos):
    """LITERAL pose_utils.h:172-252 (+144-169, 491-506) in the three null-vector sign conventions:
    index 0 raw, 1 w >= 0, 2 w <= 0  ->  (R[3,3,3], t[3,3], votes[3,4], cand[3])."""
    E = f64(E).ravel()
    c = f64(corr_aos).reshape(-1, 4)
    R, t = np.zeros((3, 9)), np.zeros((3, 3))
    votes, cand = np.zeros((3, 4), np.uint32), np.zeros(3, np.uint32)
    lib().pgo_ref_pose_from_essential(_p(E), _p(c), C.c_uint32(len(c)), _p(R), _p(t), _p(votes), _p(cand))
    return R.reshape(3, 3, 3), t, votes, cand


def ref_linear_triangulation(P1, P2, pt):
    X = np.zeros(4)
    lib().pgo_ref_linear_triangulation(_p(f64(P1).ravel()), _p(f64(P2).ravel()), _p(f64(pt).ravel()), _p(X))
    return X


def ref_decompose_essential(E):
    R1, R2, t = np.zeros(9), np.zeros(9), np.zeros(3)
    lib().pgo_ref_decompose_essential(_p(f64(E).ravel()), _p(R1), _p(R2), _p(t))
    return R1.reshape(3, 3), R2.reshape(3, 3), t


def candidate_agreement_batch(x1, y1, x2, y2, offsets, edges, masks, t_gt=None, threads=0):
    """uint16 flags per pair (oracle/pgi_oracle.c: pgo_candidate_agreement_batch)."""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    P = len(offsets) - 1
    flags = np.zeros(P, np.uint16)
    g = f64(t_gt).reshape(P, 3) if t_gt is not None else None
    lib().pgo_candidate_agreement_batch(_p(x1), _p(y1), _p(x2), _p(y2), _p(offsets), C.c_uint32(P), _p(edges), _p(masks),
                                        _p(g), _p(flags), C.c_int(threads))
    return flags


def ref_guided_match_binned(F, kp1, kp2, d1, d2, size_src, size_dst, n_bins=45):
    """LITERAL matcher.h:199-405 with the epipolar bins -> (src idx, dst idx, adapted ratio, fragile[n1])."""
    kp1, kp2, d1, d2 = f32(kp1), f32(kp2), f32(d1), f32(d2)
    n1, n2 = len(kp1), len(kp2)
    oi, oj, orr = np.zeros(max(n1, 1), np.uint32), np.zeros(max(n1, 1), np.uint32), np.zeros(max(n1, 1))
    frag = np.zeros(max(n1, 1), np.uint8)
    ss, sd = np.array(size_src, np.int32), np.array(size_dst, np.int32)
    lib().pgo_ref_guided_match_binned.restype = C.c_uint32
    m = lib().pgo_ref_guided_match_binned(_p(f64(F).ravel()), _p(kp1), C.c_uint32(n1), _p(kp2), C.c_uint32(n2), _p(d1), _p(d2),
                                          C.c_uint32(d1.shape[1] if n1 else 128), _p(ss), _p(sd), C.c_int(n_bins), _p(oi), _p(oj),
                                          _p(orr), _p(frag))
    return oi[:m], oj[:m], orr[:m], frag[:n1]


def model_from_essential(E):
    """f64 E -> unit-norm f32 model exactly as K2 (pgi_score_pose_batch) prepares it."""
    out = np.zeros(9, np.float32)
    lib().pgo_model_from_essential(_p(f64(E).ravel()), _p(out))
    return out
