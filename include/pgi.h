/*
 * pgi.h -- C ABI of the MI355X-native pairwise relative-pose engine.
 *
 * Drop-in boundary for the hot path of danini/pose-graph-initialization
 * (paths relative to /root/reference/src/pyposegraphbuilder/include/):
 *   PoseGraphBuilder::estimatePose      pose_graph_builder.h:940-1078 (decl :153-164)
 *   EssentialMatrixEvaluator::getInliers graph_traversal.h:136-168
 *   InTraversalPoseTester::test          graph_traversal.h:194-233
 *   pose::getPoseFromEssentialMatrix     pose_utils.h:172-252
 * plus the rotation averaging that BASELINE.json:north_star adds downstream.
 *
 * Plain pointers and sizes only; all buffers are caller-owned; no exceptions
 * cross the boundary; every entry point returns PGI_SUCCESS (0) or a negative
 * pgi_error and records a message readable through pgi_last_error().  The
 * library needs a HIP device: pgi_create fails (returns NULL) without one --
 * there is no CPU fallback.
 *
 * Pointers named d_* are DEVICE pointers (HBM); h_* are host pointers.
 */
#ifndef PGI_H
#define PGI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PGI_VERSION 2

typedef enum {
    PGI_SUCCESS = 0,
    PGI_ERR_INVALID = -1,  /* bad argument                                   */
    PGI_ERR_DEVICE = -2,   /* HIP runtime error (message in pgi_last_error)  */
    PGI_ERR_NOMEM = -3,
    PGI_ERR_TOO_LARGE = -4, /* a pair exceeds the supported correspondence count */
    PGI_ERR_COMM = -5       /* the multi-GPU exchange failed (RCCL error / callback error) */
} pgi_error;

/* per-edge status (pgi_edge.status) -- estimatePose's bool, refined */
#define PGI_EDGE_OK 1              /* true  (pose_graph_builder.h:1077)                  */
#define PGI_EDGE_FEW_INLIERS 0     /* false (pose_graph_builder.h:1053-1054)             */
#define PGI_EDGE_NAN (-1)          /* false (pose_graph_builder.h:1069-1070)             */
#define PGI_EDGE_FEW_POINTS (-2)   /* fewer than 5 rows                                  */
#define PGI_EDGE_TOO_MANY_ROWS (-3) /* the pair has more rows than pgi_batch.max_corr promised */

typedef struct pgi_ctx pgi_ctx;

/* Estimator parameters.  Defaults (pgi_default_params) mirror the reference's
 * call sites: confidence 0.99 (pose_graph_builder.h:1018,1042), max 1000
 * iterations (OpenCV's default for that overload), kMinimumInlierNumber 20
 * (examples/cpp_example.cpp), MSAC multiplier 3/2 (pose_graph_builder.h:963). */
typedef struct {
    double confidence;
    uint32_t max_iters;
    uint32_t round_size;    /* hypotheses per round (LO / termination granularity) */
    uint32_t lo_iters;      /* n-point refits per improvement                      */
    uint32_t min_inliers;
    uint32_t fixed_budget;  /* 0 = adaptive; else exactly this many hypotheses     */
    uint32_t guess_quirk;   /* 1 = reproduce graph_traversal.h:164 (s^2 < 1.5*thr) */
    uint32_t vote_all_rows; /* 1 = cheirality vote over all rows (pose_utils.h:203) */
    uint32_t guess_mode;    /* what a pose guess is used for.  0 = the reference's path: score the chained pose,
                               refit its inliers (pose_graph_builder.h:974-1029).  1 = rotation-guided (BASELINE
                               config 5): keep the guess's ROTATION (the only metrically meaningful
                               part of a chained pose), re-estimate the translation direction from 32 two-point
                               hypotheses (t . (p2 x R p1) = 0), local optimisation, accept at min_inliers,
                               otherwise the full robust fit */
    uint32_t lo_linear_pct; /* local optimisation refits an inlier set of at least this many percent of the rows
                               LINEARLY (smallest eigenvector of the 9x9 normal matrix = least-squares epipolar
                               matrix): as accurate as the n-point Nister refit there and several times cheaper;
                               smaller sets keep the Nister refit.  Default 35; 0 = always Nister.  A fitted model
                               that comes from a linear refit is not exactly rank 2; R and t come from its SVD
                               (pose_utils.h:144-169) like for any other model, and pgi_edge.E is rebuilt from them */
    uint32_t sampler;       /* 0 = uniform 5-row samples (default).  1 = progressive ("optionally PROSAC --
                               matches are already sorted by SNN ratio", feature_utils.h:184-186): hypothesis h draws its
                               five rows from the FIRST n(h) rows only, n(h) = max(5, floor(N * (s / 64)^(1/5))),
                               s = ceil(64 (h + 1) / max_iters) -- PROSAC's growth (n / N)^5 ~ t / T_N with T_N = max_iters
                               in 64 steps, without its forced newest point; scoring, local optimisation and the
                               stopping rule use all rows, so the gain is an earlier good model (it shows where the
                               iteration cap binds), not a lower adaptive count */
    uint32_t lo_graph_cut;  /* 0 (default) = local optimisation refits the rows inside the threshold.  lambda * 64 > 0 (9 ~ the
                               paper's 0.14) = GRAPH-CUT local optimisation, the "GC" of GC-RANSAC (Barath & Matas 2018; the
                               reference keeps only a commented-out binding, bindings.cpp:5,228): the refit's rows are the
                               minimum s-t cut of the spatial-coherence energy over the grid neighbourhood of the 4-D
                               correspondences -- kernel K = max(0, 1 - d^2 / (1.5 thr)^2) in 16 levels, cells of 1/8 per
                               axis in normalised coordinates, the rows of a cell linked in index order, cut exactly by a
                               forward / backward sweep along the chains (oracle/pgi_oracle.c: pgo_gc_labels).  Scoring, the
                               acceptance test of a refit and the stopping rule are unchanged */
} pgi_params;

/* One pose-graph edge: what estimatePose returns (SE3 + inlier count) plus E.
 * 200 bytes, identical on host and device. */
typedef struct {
    double E[9];       /* status PGI_EDGE_OK: the essential matrix OF THE RETURNED POSE, [t]x R (pose_utils.h:74-86)
                          row-major at unit Frobenius norm, with the fitted model's sign -- rank 2 with two equal
                          singular values to 1e-15, as cv::findEssentialMat's E is in the reference
                          (pose_graph_builder.h:1057-1066); the inlier mask and n_inl belong to the FITTED model, whose
                          decomposition R, t are.  Any other status: the fitted model (f32 values), if there is one */
    double R[9];       /* R_dst_src row-major   (Sophus::SE3d rotation, :1073-1075)  */
    double t[3];       /* unit t_dst_src                                            */
    int32_t status;    /* PGI_EDGE_*                                                */
    uint32_t n_inl;    /* inlierNumber_ (pose_graph_builder.h:1022,1047)            */
    uint32_t score;    /* multi-level inlier score of the final model               */
    uint32_t iters;    /* hypotheses drawn                                          */
    uint32_t votes;    /* cheirality votes of the chosen candidate (pose_utils.h:251) */
    uint32_t cand;     /* candidate index 0..3 (pose_utils.h:182,201)               */
    uint32_t used_guess;
    uint32_t lo_runs;
} pgi_edge;

/* Flattened (pair, corr) SoA batch resident in HBM.  Pair p owns rows
 * [offsets[p], offsets[p+1]).  Coordinates are normalised image coordinates as
 * produced by createCorrespondenceMatrix (pose_graph_builder.h:917-931). */
typedef struct {
    const float* d_x1;
    const float* d_y1;
    const float* d_x2;
    const float* d_y2;
    const uint64_t* d_offsets;  /* n_pairs + 1                                       */
    const double* d_thr;        /* n_pairs, normalised threshold (:934-937)          */
    const double* d_guess_Rt;   /* n_pairs*12 (R row-major, t) or NULL               */
    const uint8_t* d_has_guess; /* n_pairs or NULL                                   */
    uint32_t n_pairs;
    uint32_t max_corr;          /* max rows of any pair (sizes the LDS staging)      */
    uint64_t pair_id_base;      /* global id of pair 0 (sharding keeps seeds stable) */
    uint64_t seed;
} pgi_batch;

const char* pgi_last_error(void);
int pgi_device_count(void);
void pgi_default_params(pgi_params* p);

/* device < 0: current device.  stream: a hipStream_t (e.g. torch's current
 * stream) or NULL for the default stream; all work is enqueued there.
 * pgi_set_stream orders the new stream after everything already enqueued on the
 * previous one (event wait), because the asynchronous entry points share
 * context-owned scratch.  pgi_set_params / pgi_set_stream are thread-safe. */
pgi_ctx* pgi_create(int device, const pgi_params* params);
void pgi_destroy(pgi_ctx* ctx);
int pgi_set_stream(pgi_ctx* ctx, void* hip_stream);
/* The stream the context enqueues on right now (NULL = the default stream): what a caller that feeds the context from
 * its own copy stream must order its events against (hipStreamWaitEvent(*out, ev)). */
int pgi_get_stream(pgi_ctx* ctx, void** hip_stream_out);
/* The HIP device ordinal the context was created on (pgi_create's `device`, resolved when it was < 0). */
int pgi_get_device(pgi_ctx* ctx, int* device_out);
int pgi_set_params(pgi_ctx* ctx, const pgi_params* params);
int pgi_get_params(pgi_ctx* ctx, pgi_params* params);
int pgi_synchronize(pgi_ctx* ctx);

/* ---- estimatePose, batched (asynchronous on the ctx stream) ------------- */
/* d_edges: n_pairs records; d_masks: one byte per row of the batch. */
int pgi_estimate_pose_batch(pgi_ctx* ctx, const pgi_batch* batch, pgi_edge* d_edges,
                            uint8_t* d_masks);
/* Host buffers in, host buffers out (synchronous).  Same result as pgi_estimate_pose_batch on the same rows, ids and
 * seed; internally the batch travels in chunks (a small first one, then multiples of the number of workgroups the
 * device keeps resident) through four device slots, two alternating kernel streams and one high-priority copy stream,
 * so PCIe copies overlap the kernels and the tail of one chunk's kernel overlaps the head of the next.  h_guess_Rt / h_has_guess may both be NULL.
 * h_masks: one byte per row.  Fastest with page-locked buffers (pgi_host_register / hipHostMalloc): when the four
 * coordinate arrays, h_edges and h_masks are all page-locked the kernel works on them in place over PCIe (each row is read
 * once, each result written once; no copy through HBM).  Same results either way. */
int pgi_estimate_pose_batch_host(pgi_ctx* ctx, const float* h_x1, const float* h_y1, const float* h_x2,
                                 const float* h_y2, const uint64_t* h_offsets, const double* h_thr,
                                 const double* h_guess_Rt, const uint8_t* h_has_guess, uint32_t n_pairs,
                                 uint64_t pair_id_base, uint64_t seed, pgi_edge* h_edges, uint8_t* h_masks);

/* Page-locks a caller-owned host buffer (hipHostRegister) so that the host-pointer entry points move it by true
 * asynchronous DMA at PCIe rate instead of through the runtime's pageable staging path; buffers that are not
 * registered keep working, slower.  The registration also maps the buffer into the device's address space (in-place
 * use by pgi_estimate_pose_batch_host).  Register once, reuse across calls; unregister before freeing the memory. */
int pgi_host_register(void* h_ptr, uint64_t bytes);
int pgi_host_unregister(void* h_ptr);

/* ---- estimatePose, literal drop-in (host pointers, synchronous, re-entrant)
 * corr_aos: n x 4 doubles [x1 y1 x2 y2] == cv::Mat N x 4 CV_64F (:946);
 * guesses_Rt: g x 12 doubles (the reference passes 0 or 1 guess, SURVEY §8a-12;
 * the LAST guess wins as in :974-1029); min_inliers: the seam's own
 * kMinimumInlierNumber_ argument (:155; 0 = the context's parameter), applied
 * to this call only; mask: n bytes (std::vector<uchar>).
 * Re-entrant like the reference's seam, which is called from kCoreNumber OpenMP
 * threads (:391-392): every call takes a private slot (own stream, device
 * scratch and pinned staging) from a pool of PGI_PAIR_SLOTS, so concurrent
 * callers overlap on the GPU instead of queueing behind one another.
 * Returns 1 (true), 0 (false) or a negative pgi_error. */
#define PGI_PAIR_SLOTS 32
int pgi_estimate_pose(pgi_ctx* ctx, const double* h_corr_aos, uint32_t n, double thr,
                      const double* h_guesses_Rt, uint32_t g, uint32_t min_inliers, uint64_t seed,
                      uint64_t pair_id, pgi_edge* h_edge, uint8_t* h_mask);

/* ---- getInliers / InTraversalPoseTester, batched ------------------------ */
/* One model per pair.  d_E: n_pairs x 9 doubles.  d_tau2: per-pair bound on the
 * SQUARED Sampson distance (pass 1.5*thr for the graph_traversal.h:164 quirk,
 * (1.5*thr)^2 for :184).  d_masks may be NULL.  f32 arithmetic on the SoA batch. */
int pgi_score_pose_batch(pgi_ctx* ctx, const pgi_batch* batch, const double* d_E,
                         const double* d_tau2, uint32_t* d_counts, uint8_t* d_masks);
/* Reference-layout variant: f64 AoS rows (n x 4) and the reference's exact
 * operation order (graph_traversal.h:107-115); bit-identical to it. */
int pgi_score_pose_f64(pgi_ctx* ctx, const double* d_corr_aos, const uint64_t* d_offsets,
                       uint32_t n_pairs, const double* d_E, const double* d_tau2,
                       uint32_t* d_counts, uint8_t* d_masks);

/* The tester's decision for a whole batch without a host round trip: d_has_guess[p] is cleared where d_counts[p] (from
 * pgi_score_pose_batch with tau2 = (1.5 thr)^2) is below min_count -- InTraversalPoseTester::test accepts a chained pose at
 * kMinimumInlierNumber = 5 inliers (pose_graph_builder.h:809, graph_traversal.h:221-225).  Asynchronous on the stream. */
int pgi_screen_guesses(pgi_ctx* ctx, const uint32_t* d_counts, uint32_t min_count, uint8_t* d_has_guess, uint32_t n_pairs);

/* The two scoring seams A* calls per tested path, for ONE pair with HOST pointers (re-entrant: each call leases
 * one of the PGI_PAIR_SLOTS private slots -- stream, device scratch, pinned staging -- that pgi_estimate_pose
 * uses, so the reference's 20 traversal threads overlap on the GPU; no allocation per call once a slot is warm):
 *   EssentialMatrixEvaluator::getInliers (graph_traversal.h:136-168): early_exit_at = 0, h_mask = n bytes,
 *       tau2 = kThreshold_ as passed (the reference compares the SQUARED residual with it, :164);
 *   InTraversalPoseTester::test (graph_traversal.h:194-233): early_exit_at = kMinimumInlierNumber_, h_mask = NULL,
 *       tau2 = the squared threshold (:184); the scan stops after the 256-row chunk holding that inlier.
 * h_corr_aos: n x 4 doubles (cv::Mat N x 4 CV_64F); E: row-major 3 x 3; same arithmetic and operation order as
 * pgi_score_pose_f64.  Returns 1 when early_exit_at > 0 was reached (*count = early_exit_at: the reference returns
 * AT that inlier, :221-225), 0 otherwise (*count = rows that pass), negative on error.  h_mask may be NULL. */
int pgi_score_pose_f64_host(pgi_ctx* ctx, const double* h_corr_aos, uint32_t n, const double E[9], double tau2,
                            uint32_t early_exit_at, uint32_t* count, uint8_t* h_mask);

/* ---- getPoseFromEssentialMatrix, batched -------------------------------- */
/* d_E: n_pairs x 9; d_masks: rows voting (NULL, or pgi_params.vote_all_rows = 1:
 * all rows).  Writes R, t, votes, cand of d_edges (other fields untouched). */
int pgi_decompose_batch(pgi_ctx* ctx, const pgi_batch* batch, const double* d_E,
                        const uint8_t* d_masks, pgi_edge* d_edges);

/* pose::getPoseFromEssentialMatrix (pose_utils.h:172-252) as the reference calls it -- ONE pair, HOST pointers: E (row-major
 * 3 x 3) and the N x 4 CV_64F correspondence matrix in, rotation (row-major), unit translation and the winning candidate's
 * vote count out (the reference's return value, :251).  Re-entrant through the PGI_PAIR_SLOTS pool like pgi_estimate_pose.
 * h_mask (n bytes, may be NULL): the rows that vote; NULL, or pgi_params.vote_all_rows = 1: every row, the reference's
 * population (:203).  Candidate order, first-maximum rule and the sign rule t_out = (cand odd ? -t : +t) are the
 * reference's (:182, :201, :243-250); the per-row test is the depth-sign rule of the product (same
 * rotation as the reference's 4 x 4 DLT + raw-z test in 100 % of 102 000 measured pairs).  cand (optional): 0..3. */
int pgi_pose_from_essential_host(pgi_ctx* ctx, const double E[9], const double* h_corr_aos, uint32_t n, const uint8_t* h_mask,
                                 double R[9], double t[3], uint32_t* votes, uint32_t* cand);

/* ---- minimal solver, batched (5-point; debugging / parity) -------------- */
/* d_pts: n_samples x 5 x 4 floats.  d_models: n_samples x 10 x 9 floats,
 * d_counts: n_samples.  d_dbg (optional): n_samples x PGI_DBG_DOUBLES doubles:
 * basis[36] cons[200] red[100] poly[11] roots[10] nroots[1]. */
#define PGI_DBG_DOUBLES 358
int pgi_five_point_batch(pgi_ctx* ctx, const float* d_pts, uint32_t n_samples, float* d_models,
                         uint32_t* d_counts, double* d_dbg);

/* ---- rotation averaging (north_star: L1 + IRLS; not in the reference) ---- */
typedef struct {
    uint32_t src, dst;  /* R_rel ~ R_dst * R_src^T  (T_dst_src, pose.h:14)          */
    double R[9];
    double weight;      /* edge score = inlier ratio (pose_graph_builder.h:645-654) */
} pgi_rot_edge;
typedef struct {
    uint32_t l1_iters;     /* 5    */
    uint32_t irls_iters;   /* 100  */
    uint32_t cg_iters;     /* 200  */
    double sigma_deg;      /* 5.0  */
    double tol;            /* 1e-8 mean step, radians */
} pgi_rotavg_params;
void pgi_default_rotavg_params(pgi_rotavg_params* p);
/* h_* host pointers; R_out: n_views x 9 (world->camera), view 0 of each
 * connected component fixed to its spanning-tree value.  Synchronous. */
int pgi_rotation_average(pgi_ctx* ctx, const pgi_rot_edge* h_edges, uint32_t n_edges,
                         uint32_t n_views, const pgi_rotavg_params* prm, double* h_R_out,
                         uint32_t* h_iters_out);

/* The same solve fed from the DEVICE-resident edge table the all-gather leaves on every rank (no
 * 200-byte-per-edge host round trip): pair p = (h_src[p], h_dst[p]) with record d_edges[p]; records
 * whose status is not PGI_EDGE_OK are skipped; weight = n_inl / h_rows[p] (the reference's edge score,
 * pose_graph_builder.h:645-654; h_rows NULL = weight 1).  Only status/n_inl (8 B per pair) and the
 * rotations of the spanning-forest edges travel to the host. */
int pgi_rotation_average_edges(pgi_ctx* ctx, const pgi_edge* d_edges, const uint32_t* h_src, const uint32_t* h_dst,
                               const uint32_t* h_rows, uint32_t n_pairs, uint32_t n_views,
                               const pgi_rotavg_params* prm, double* h_R_out, uint32_t* h_iters_out,
                               uint32_t* h_edges_used);

/* ---- multi-GPU: the path's single exchange step -----------------------------------
 * Pairs are sharded over ranks (one process per GPU); every rank estimates its own block and the
 * fixed-size edge records are all-gathered so that each rank holds the full table for the
 * replicated rotation averaging / the next scheduler wave.  Two transports:
 *   RCCL  (production: one GPU per rank, xGMI): rank 0 calls pgi_comm_unique_id, the caller ships the
 *         128 bytes to the other ranks (any bootstrap: TCP, torch.distributed, MPI), every rank calls
 *         pgi_comm_init_rccl.  Equal blocks: one ncclAllGather.  Uneven blocks are gathered without padding
 *         as one group of point-to-point ncclSend / ncclRecv (every rank pushes its block to its world-1 peers
 *         over world-1 different xGMI links) on the context stream (asynchronous).  pgi_comm_init_rccl is
 *         a collective: call pgi_comm_rccl_probe on every rank first and agree on the result, so that no rank
 *         enters it alone (host/distributed.cpp attach(), pyposegraphbuilder.distributed.Communicator do).
 *   host  (ranks sharing a device -- RCCL refuses that -- and CPU-side tests): the caller supplies an
 *         all-gather-v over host memory; the records make a D2H / H2D hop (synchronous).
 * Without a communicator (single process) pgi_allgather_edges degenerates to a device copy. */
#define PGI_COMM_ID_BYTES 128
/* gathers send_bytes from every rank into recv (rank order); recv_bytes[r] = bytes of rank r; 0 = ok */
typedef int (*pgi_allgatherv_fn)(void* user, const void* h_send, uint64_t send_bytes, void* h_recv,
                                 const uint64_t* recv_bytes, uint32_t world);
/* 0 iff librccl could be opened and carries every entry point this library needs (no collective, no device
 * call); *version (optional) = ncclGetVersion's code, e.g. 22606, or 0 when unknown */
int pgi_comm_rccl_probe(int* version);
int pgi_comm_unique_id(uint8_t id[PGI_COMM_ID_BYTES]);
int pgi_comm_init_rccl(pgi_ctx* ctx, uint32_t world, uint32_t rank, const uint8_t id[PGI_COMM_ID_BYTES]);
int pgi_comm_init_host(pgi_ctx* ctx, uint32_t world, uint32_t rank, pgi_allgatherv_fn fn, void* user);
int pgi_comm_destroy(pgi_ctx* ctx);
/* world = 1, rank = 0 without a communicator; *kind: 0 none, 1 RCCL, 2 host */
int pgi_comm_info(pgi_ctx* ctx, uint32_t* world, uint32_t* rank, uint32_t* kind);
/* h_counts[world]: records contributed by each rank (h_counts[rank] = this rank's; the partition is
 * known to every rank by construction, so no size exchange is needed).  d_all receives sum(h_counts)
 * records in rank order == global pair order for contiguous blocks.  d_local may alias its own slot
 * of d_all. */
int pgi_allgather_edges(pgi_ctx* ctx, const pgi_edge* d_local, const uint32_t* h_counts, pgi_edge* d_all);
/* the same for raw bytes (inlier masks, match lists): h_bytes[world] */
int pgi_allgatherv(pgi_ctx* ctx, const void* d_local, const uint64_t* h_bytes, void* d_all);

/* ---- descriptor matching (feature_utils.h:135-202) ----------
 * The step that produces the correspondences estimatePose consumes: two brute-force
 * L2 kNN(2) searches (cv::BFMatcher, :151-163), Lowe ratio 0.90 and mutual-best test
 * (:165-176), matches sorted by the ratio (:178-180).  Descriptors are PGI_DESC_DIM
 * floats (RootSIFT, feature_utils.h:26-133).  Distances come from the exact-f32 MFMA
 * (v_mfma_f32_32x32x2_f32 == an fmaf chain over k = 0..127), so the result is
 * bit-identical to the scalar specification in oracle/pgi_oracle.c. */
#define PGI_DESC_DIM 128
#define PGI_DESC_MAX 16384 /* keypoints per image supported by the selection kernel */
/* rounds n up to the padded keypoint count used by the transposed layout */
uint32_t pgi_desc_padded(uint32_t n);
/* d_desc: n x 128 row-major floats (cv::Mat descriptors).  Writes the transposed copy
 * d_desc_t (128 x pgi_desc_padded(n), zero padded) and the squared norms d_norm
 * (pgi_desc_padded(n)).  Done once per image; every pair the image takes part in reuses it. */
int pgi_desc_prepare(pgi_ctx* ctx, const float* d_desc, uint32_t n, float* d_desc_t, float* d_norm);
typedef struct {
    const float* d_desc_t; /* 128 x n_pad */
    const float* d_norm;   /* n_pad       */
    uint32_t n;
    uint32_t n_pad;        /* == pgi_desc_padded(n) */
    /* optional (pgi_desc_prepare_screen): when every image of a batch carries them, the matcher screens with
     * f16 matrix-core products (16x the f32 rate) and re-evaluates only the few candidates that can matter
     * with the exact f32 chain -- same output, bit for bit; NULL selects the all-f32 kernel */
    const float* d_desc_rm;       /* n_pad x 128 row-major f32, zero padded */
    const uint16_t* d_desc_f16;   /* n_pad x 128 IEEE halves (round to nearest even) in the matcher's own tile order:
                                     opaque, written by pgi_desc_prepare_screen */
} pgi_desc_view;
/* d_desc: n x 128 row-major floats -> the padded row-major copy (n_pad x 128 floats) and its half-precision rounding
 * (n_pad x 128 halves, tile-ordered for the matrix cores). */
int pgi_desc_prepare_screen(pgi_ctx* ctx, const float* d_desc, uint32_t n, float* d_desc_rm, uint16_t* d_desc_f16);
/* Pair p matches image h_src[p] (queries) against h_dst[p].  Outputs, per pair p at
 * stride max_matches: d_match_src / d_match_dst (keypoint indices, :183-197) and d_ratio,
 * sorted by (ratio, src index); d_counts[p] = number of matches.  Asynchronous on the stream. */
int pgi_match_descriptors_batch(pgi_ctx* ctx, const pgi_desc_view* h_src, const pgi_desc_view* h_dst,
                                uint32_t n_pairs, uint32_t max_matches, uint32_t* d_match_src,
                                uint32_t* d_match_dst, double* d_ratio, uint32_t* d_counts);

/* ---- createCorrespondenceMatrix on the device (pose_graph_builder.h:864-938) ------------
 * Turns the matcher's output into the pgi_batch the estimator consumes, without a host round
 * trip: gathers cv::KeyPoint::pt of every match, normalises with the pinhole intrinsics
 * K = [fx 0 cx; 0 fy cy; 0 0 1] (:286) in double, rounds to f32, normalises the threshold
 * (:934-937) and writes the row offsets (exclusive scan of the kept counts). */
typedef struct {
    const float* d_xy; /* n x 2 pixel coordinates (x, y) */
    uint32_t n;
    uint32_t reserved;
    double fx, fy, cx, cy;
} pgi_keypoint_view;
/* top_k: keep only the first top_k matches of a pair (0 = all; the tracklet path keeps 100,
 * pose_graph_builder.h:759-772).  dst_uses_src_intrinsics = 1 reproduces :908-912.
 * d_x1..d_y2 need room for n_pairs * min(max_matches, top_k) floats each; d_offsets n_pairs + 1;
 * d_thr n_pairs.  Asynchronous on the stream. */
int pgi_build_correspondences(pgi_ctx* ctx, const pgi_keypoint_view* h_src, const pgi_keypoint_view* h_dst,
                              uint32_t n_pairs, uint32_t max_matches, const uint32_t* d_match_src,
                              const uint32_t* d_match_dst, const uint32_t* d_counts, uint32_t top_k,
                              double thr_px, uint32_t dst_uses_src_intrinsics, float* d_x1, float* d_y1,
                              float* d_x2, float* d_y2, uint64_t* d_offsets, double* d_thr);

/* ---- guided matching with a known pose (matcher.h:199-405, ---------------
 * pose_graph_builder.h:715-783).  After estimatePose succeeded on tracklet correspondences the
 * reference looks for additional matches consistent with the pose: for every source keypoint,
 * the destination keypoints within 0.75 px symmetric epipolar distance (:340-354) compete by
 * descriptor SSD, and the winner is kept if the count-adapted ratio test passes (:375-399).
 * n_bins > 0 (the reference instantiates 45, pose_graph_builder.h:738) reproduces its epipolar
 * hashing: destination keypoints are binned by the angle of their epipolar line's normal in the
 * source image, and a source keypoint only competes inside its own bin (matcher.h:218-331).
 * n_bins = 0 visits every destination keypoint (the loop the reference keeps commented at :327) --
 * a superset of the binned candidates, so ~1 % of the matches differ from the reference's.
 * If more than max_n matches survive, the max_n with the smallest adapted ratio are returned in
 * that order (pose_graph_builder.h:759-772); otherwise all, in source order.  max_n = 0 keeps
 * everything.
 * Two implementations of the binned mode give the same bytes (tests/test_guided.py, scripts/soak_guided.py): the
 * default sorts the destination records by a fine angle bucket and evaluates descriptor rows staged once per
 * wavefront (guided_scan_tile_kernel; images of up to 65535 keypoints), PGI_GUIDED_ANGLE=0 in the environment
 * selects the scan over whole bins; PGI_GUIDED_LANES = 1 | 2 | 4 (default 2) sets the lanes per source keypoint
 * of the former.  Both variables are read per call and exist for A/B measurements only. */
typedef struct {
    const float* d_xy;   /* n x 2 pixel coordinates        */
    const float* d_desc; /* n x 128 row-major descriptors  */
    uint32_t n;
    uint32_t reserved;
    double fx, fy, cx, cy;
    double width, height; /* image size (PinholeCamera::getWidth/getHeight -> cv::Size, pose_graph_builder.h:741-742);
                             read only when n_bins > 0 */
} pgi_feature_view;
/* h_pose_Rt: n_pairs x 12 doubles (R_dst_src row-major, t_dst_src).  Outputs per pair at stride
 * out_stride (>= max_n, or >= the largest source keypoint count when max_n = 0).  d_ratio holds
 * dist_ratio_sq_adapted (:394).  Asynchronous on the stream.  Workspace (grow-only, owned by the context, shared with the
 * descriptor matcher): ~100 bytes per keypoint of the batch plus, for n_bins > 0, 10 KB per 64 source keypoints (the lists the
 * scan's first kernel hands to its second: 0.7 GB for 512 pairs x 8000 keypoints; the first call of a larger batch allocates). */
int pgi_guided_match_batch(pgi_ctx* ctx, const pgi_feature_view* h_src, const pgi_feature_view* h_dst,
                           uint32_t n_pairs, const double* h_pose_Rt, uint32_t n_bins, uint32_t max_n,
                           uint32_t out_stride, uint32_t* d_match_src, uint32_t* d_match_dst, double* d_ratio,
                           uint32_t* d_counts);

/* ---- tracklets resident in HBM (reconstruction::Tracklets, point_track.h:541-711) ----
 * The store behind "quick matching": add() grows multi-view tracks from the inlier matches of an
 * estimated edge (:633-711); getCorrespondences() returns, for a later pair, the keypoints both views
 * share a track with (:568-631).  Results are those of the reference applied one match at a time, in
 * array order -- track indices, member order, the order of every view's track list and the reference's
 * quirks included (oracle/tracklets_oracle.py) -- but the matches of a batch run concurrently wherever
 * they do not share a keypoint.  View indices are < n_views; keypoint indices are any
 * 32-bit value. */
typedef struct pgi_tracklets pgi_tracklets;
pgi_tracklets* pgi_tracklets_create(pgi_ctx* ctx, uint32_t n_views); /* NULL on failure (pgi_last_error) */
void pgi_tracklets_destroy(pgi_tracklets* trk);                       /* before pgi_destroy of its context */
/* One add() call of the reference (point_track.h:633-636): matches (d_src[i], d_dst[i]), i < *d_count
 * (n_max when d_count is NULL), kept where d_mask[i] != 0 (all when d_mask is NULL).  Device pointers;
 * the arrays are read while the call runs. */
typedef struct {
    uint32_t view_src, view_dst; /* imageIdxSource_, imageIdxDestination_ (must differ) */
    uint32_t n_max;              /* capacity of the arrays                               */
    uint32_t reserved;
    const uint32_t* d_src;       /* keypoint index in the source view      */
    const uint32_t* d_dst;       /* keypoint index in the destination view */
    const uint8_t* d_mask;       /* inlierMask_, or NULL                   */
    const uint32_t* d_count;     /* number of matches on the device, or NULL */
} pgi_tracklet_pair;
/* Applies h_pairs[0..n_pairs) in order.  Synchronous: the store is consistent on return. */
int pgi_tracklets_add_batch(pgi_tracklets* trk, const pgi_tracklet_pair* h_pairs, uint32_t n_pairs);
/* getCorrespondences for n_queries (source, destination) pairs: query q writes d_count[q] <= max_n + 1
 * entries (the reference stops only after exceeding the maximum, :626-627) to d_src_idx / d_dst_idx
 * at q * out_stride, in the order of the destination view's track list.  out_stride >= max_n + 1.
 * Asynchronous on the stream. */
int pgi_tracklets_get_batch(pgi_tracklets* trk, const uint32_t* h_view_src, const uint32_t* h_view_dst,
                            uint32_t n_queries, uint32_t max_n, uint32_t out_stride, uint32_t* d_src_idx,
                            uint32_t* d_dst_idx, uint32_t* d_count);
/* Tracks, events (members over all tracks) and the number of launches the last add_batch needed. */
int pgi_tracklets_info(const pgi_tracklets* trk, uint64_t* n_tracks, uint64_t* n_events, uint32_t* last_rounds);
/* Members of one track in insertion order as (view << 32 | keypoint); *n_members is the full count,
 * at most `capacity` are written to the HOST array h_members. */
int pgi_tracklets_track(pgi_tracklets* trk, uint64_t index, uint64_t* h_members, uint32_t capacity,
                        uint32_t* n_members);

#ifdef __cplusplus
}
#endif
#endif
