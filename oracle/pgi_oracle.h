/*
 * pgi_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the pairwise relative-pose hot path of
 * danini/pose-graph-initialization.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped HIP path in
 * pose-graph-initialization_amd/ never includes, links or calls anything here.
 *
 * PARITY UNPINNED against OpenCV: the reference delegates the robust estimator
 * to cv::findEssentialMat (pose_graph_builder.h:1013-1020, 1037-1044) -- OpenCV,
 * version un-pinned (CMakeLists.txt:26), absent from /root/reference and from
 * this image, and the reference holds no tests or golden vectors.
 * What IS pinned: the in-tree arithmetic the reference does itself
 * (graph_traversal.h:86-233, pose_utils.h:74-252, pose.h:46-98,
 * pose_graph_builder.h:940-1078 control flow) is restated function-by-function
 * below with file:line citations and checked against independent numpy/scipy
 * arithmetic in tests/ (tests/golden/make_golden.py generates the fixtures).
 *
 * The estimator slot (5-point minimal solver + multi-level Sampson inlier
 * scoring + n-point local-optimisation refit) follows the published
 * algorithms named in BASELINE.json:north_star (Nister 2004 five-point;
 * GC-RANSAC lineage LO), specified op-by-op in DESIGN.md §3 so that the HIP
 * kernels reproduce it BIT-EXACTLY: every floating-point operation below is a
 * single IEEE-754 op (+,-,*,/,sqrt,fma) in a fixed order; build with
 * -ffp-contract=off.
 */
#ifndef PGI_ORACLE_H
#define PGI_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PGO_MAX_MODELS 10      /* real roots of the degree-10 polynomial          */
#ifndef PGO_GRID
#define PGO_GRID 256           /* root-bracketing intervals                       */
#endif
#ifndef PGO_NEWTON_ITERS
#define PGO_NEWTON_ITERS 10    /* safeguarded Newton iterations per bracket       */
#endif
#ifndef PGO_JACOBI9_SWEEPS
#define PGO_JACOBI9_SWEEPS 6   /* cyclic (tournament-ordered) Jacobi sweeps, 9x9  */
#endif
#define PGO_GUIDED_HYPS 32     /* two-point translation hypotheses of the rotation-guided guess path */
#ifndef PGO_SVD3_SWEEPS
#define PGO_SVD3_SWEEPS 4      /* one-sided Jacobi sweeps, 3x3                    */
#endif

/* status codes of an edge */
#define PGO_OK 1
#define PGO_FAIL_FEW_INLIERS 0
#define PGO_FAIL_NAN (-1)
#define PGO_FAIL_FEW_POINTS (-2)

typedef struct {
    double confidence;       /* 0.99  (pose_graph_builder.h:1018,1042)                   */
    uint32_t max_iters;      /* 1000  (OpenCV default of this overload)    */
    uint32_t round_size;     /* hypotheses per round (termination / LO granularity)     */
    uint32_t lo_iters;       /* max n-point refits per improvement                      */
    uint32_t min_inliers;    /* kMinimumInlierNumber = 20 (cpp_example.cpp)             */
    uint32_t fixed_budget;   /* 0 = adaptive; else exactly this many hypotheses         */
    uint32_t guess_quirk;    /* 1 = getInliers compares s^2 < 1.5*thr (graph_traversal.h:164) */
    uint32_t vote_all_rows;  /* 1 = cheirality vote over all rows (pose_utils.h:203)    */
    uint32_t guess_mode;     /* 0 = the reference's guess path (score -> all-inlier refit, :974-1029);
                                1 = rotation-guided: keep R of the guess, re-estimate t (BASELINE config 5) */
    uint32_t lo_linear_pct;  /* LO refits an inlier set of at least this many percent of the rows linearly
                                (pgo_linear_refit); smaller sets with the n-point Nister refit.  0 = always Nister */
    uint32_t sampler;        /* 0 uniform, 1 progressive prefix sampling (see include/pgi.h) */
    uint32_t lo_graph_cut;   /* 0 = LO refits the rows inside the threshold; lambda * 64 > 0 = graph-cut local optimisation:
                                the refit's rows are the minimum cut of the spatial-coherence energy (pgo_gc_labels) */
} pgo_params;

typedef struct {
    double E[9];             /* row-major, unit Frobenius norm                          */
    double R[9];             /* row-major R_dst_src                                     */
    double t[3];             /* unit t_dst_src                                          */
    int32_t status;
    uint32_t n_inl;          /* inliers at thr (sampson^2 < thr^2)                       */
    uint32_t score;          /* multi-level inlier score of the final model             */
    uint32_t iters;          /* hypotheses drawn                                        */
    uint32_t votes;          /* cheirality votes of the chosen candidate                */
    uint32_t cand;           /* chosen candidate 0..3                                   */
    uint32_t used_guess;     /* 1 if the pose guess was accepted                        */
    uint32_t lo_runs;        /* number of n-point refits executed                       */
} pgo_edge;

void pgo_default_params(pgo_params* p);

/* ---- graph-cut local optimisation (GC-RANSAC's labelling step; see pgi_oracle.c) ------------------ */
#define PGO_GC_LEVELS 16u
#define PGO_GC_UNARY 128      /* weight of one kernel level in the unary terms */
#define PGO_GC_NONE 0xFFFFFFFFu
uint32_t pgo_gc_cell(float x1, float y1, float x2, float y2);
void pgo_gc_chains(const float* x1, const float* y1, const float* x2, const float* y2, uint32_t n, uint32_t* prev);
uint32_t pgo_gc_kernel_level(const float E[9], float x1, float y1, float x2, float y2, float thr2);
uint32_t pgo_gc_labels(const float E[9], const float* x1, const float* y1, const float* x2, const float* y2, uint32_t n,
                       double thr, uint32_t lambda64, const uint32_t* prev, uint8_t* labels, int64_t* energy);
uint32_t pgo_gc_cut(const uint32_t* k, const uint32_t* prev, uint32_t n, uint32_t lambda64, uint8_t* labels);
int64_t pgo_gc_energy(const uint32_t* k, const uint32_t* prev, const uint8_t* labels, uint32_t n, uint32_t lambda64);

/* ---- reference in-tree arithmetic, f64 (restatements) ------------------ */
/* graph_traversal.h:86-116 */
double pgo_ref_sampson_sq(const double corr[4], const double E[9]);
/* graph_traversal.h:136-168 (thr compared UN-squared: the quirk) */
uint32_t pgo_ref_get_inliers(const double* corr_aos, uint32_t n, const double E[9],
                             double thr, uint32_t* idx_out);
/* graph_traversal.h:194-233 */
int pgo_ref_pose_test(const double* corr_aos, uint32_t n, const double R[9],
                      const double t[3], double thr, uint32_t min_inl, uint32_t* n_inl);
/* pose_utils.h:74-86 */
void pgo_ref_essential_from_pose(const double R[9], const double t[3], double E[9]);
/* graph_traversal.h:290-348 (one step: T <- T_edge * T or T_edge^-1 * T) */
void pgo_ref_chain_pose(const double Re[9], const double te[3], int inverted,
                        double R[9], double t[3]);
/* pose_graph_builder.h:864-938 (src_intrinsics_for_dst=1 reproduces :908-912) */
void pgo_ref_normalize_corr(const float* kp_src_xy, const float* kp_dst_xy,
                            const uint32_t* match_src, const uint32_t* match_dst, uint32_t m,
                            double f_src, double w_src, double h_src,
                            double f_dst, double w_dst, double h_dst,
                            int src_intrinsics_for_dst, double thr_px,
                            double* corr_aos, double* thr_norm);

/* ---- engine spec pieces (bit-exact targets for the HIP kernels) -------- */
uint64_t pgo_mix64(uint64_t z);
uint32_t pgo_draw_index(uint64_t base, uint32_t hyp, uint32_t k, uint32_t n);
void pgo_sample5(uint64_t seed, uint64_t pair_id, uint32_t hyp, uint32_t n, uint32_t idx[5]);

/* multi-level Sampson score of one f32 model over f32 SoA points */
void pgo_score_model(const float E[9], const float* x1, const float* y1, const float* x2,
                     const float* y2, uint32_t n, double thr, uint32_t* score,
                     uint32_t* n_inl);
/* pre-verification on the first min(64,n) rows against a best with n_bar inliers (1 = keep) */
int pgo_preverify(const float E[9], const float* x1, const float* y1, const float* x2,
                  const float* y2, uint32_t n, double thr, uint32_t n_bar);
/* f64 E -> the unit-norm f32 model the one-model scoring kernel uses (fma-chain norm) */
void pgo_model_from_essential(const double E[9], float e32[9]);
/* mask[i] = sampson^2 < tau2 (tau2 arbitrary; f32 arithmetic) */
uint32_t pgo_mask_model(const float E[9], const float* x1, const float* y1, const float* x2,
                        const float* y2, uint32_t n, float tau2, uint8_t* mask);

/* 5x9 null space (orthonormalised), basis[4][9] = X,Y,Z,W */
void pgo_nullspace5(const float pts[5][4], double basis[36]);
/* Nister back-end on a 4-vector basis: fills poly[11], roots, models (f32, unit norm) */
typedef struct {
    double cons[10][20];   /* constraint matrix before elimination          */
    double red[10][10];    /* right block after Gauss-Jordan, by pivot col  */
    double poly[11];       /* degree-10 polynomial, poly[c] * z^c            */
    double roots[PGO_MAX_MODELS];
    uint32_t n_roots;
} pgo_backend_dbg;
uint32_t pgo_backend(const double basis[36], const float (*sample)[4], uint32_t n_sample,
                     float models[PGO_MAX_MODELS][9], pgo_backend_dbg* dbg);
uint32_t pgo_five_point(const float pts[5][4], float models[PGO_MAX_MODELS][9],
                        pgo_backend_dbg* dbg);
/* quantised-exact normal matrix of the inlier set + Jacobi eigenbasis */
void pgo_normal_matrix(const float* x1, const float* y1, const float* x2, const float* y2,
                       const uint8_t* mask, uint32_t n, double A[81]);
void pgo_jacobi9(double A[81], double V[81]);
void pgo_basis_from_eigen(const double A[81], const double V[81], double basis[36]);
uint32_t pgo_npoint(const float* x1, const float* y1, const float* x2, const float* y2,
                    const uint8_t* mask, uint32_t n, float models[PGO_MAX_MODELS][9]);

uint32_t pgo_linear_refit(const float* x1, const float* y1, const float* x2, const float* y2, const uint8_t* mask,
                          uint32_t n, float model[9]);
void pgo_svd3(const double E[9], double U[9], double S[3], double V[9]);
/* pose_utils.h:144-252 structure, correct cheirality */
void pgo_decompose(const double E[9], const float* x1, const float* y1, const float* x2,
                   const float* y2, const uint8_t* mask, uint32_t n, int vote_all,
                   double R[9], double t[3], uint32_t votes[4], uint32_t* cand);

/* LITERAL restatements of the reference's candidate selection (pose_utils.h:144-169, 172-252, 491-506); used only
 * to measure how the product's depth-sign vote differs from it (scripts/candidate_agreement.py) */
void pgo_ref_linear_triangulation(const double P1[12], const double P2[12], const double pt[4], double X[4]);
void pgo_ref_decompose_essential(const double E[9], double R1[9], double R2[9], double t[3]);
int pgo_ref_pose_from_essential(const double E[9], const double* corr_aos, uint32_t n, double out_R[3][9],
                                double out_t[3][3], uint32_t out_votes[3][4], uint32_t out_cand[3]);
void pgo_candidate_agreement_batch(const float* x1, const float* y1, const float* x2, const float* y2,
                                   const uint64_t* offsets, uint32_t n_pairs, const pgo_edge* edges, const uint8_t* masks,
                                   const double* t_gt, uint16_t* flags, int threads);

/* robust fit: rounds of hypotheses, LO on improvement, adaptive termination */
void pgo_ransac_essential(const float* x1, const float* y1, const float* x2, const float* y2,
                          uint32_t n, double thr, const pgo_params* prm, uint64_t seed,
                          uint64_t pair_id, pgo_edge* out, uint8_t* mask);

/* pose_graph_builder.h:940-1078 control flow */
void pgo_estimate_pose(const float* x1, const float* y1, const float* x2, const float* y2,
                       uint32_t n, double thr, const double* guess_Rt /*12 or NULL*/,
                       const pgo_params* prm, uint64_t seed, uint64_t pair_id, pgo_edge* out,
                       uint8_t* mask);

/* mutual nearest neighbour + ratio test of two descriptor sets (feature_utils.h:135-202); returns #matches */
uint32_t pgo_match_descriptors(const float* A, uint32_t k1, const float* B, uint32_t k2, uint32_t d,
                               uint32_t* out_i, uint32_t* out_j, double* out_ratio);

/* guided matching with a known pose (matcher.h:199-405), exhaustive candidate loop; returns #matches in source order */
void pgo_fundamental_from_essential(const double E[9], const double k_src[4], const double k_dst[4], double F[9]);
uint32_t pgo_guided_match(const double F[9], const float* kp1, uint32_t n1, const float* kp2, uint32_t n2,
                          const float* d1, const float* d2, uint32_t dim, uint32_t* out_i, uint32_t* out_j,
                          double* out_ratio);

/* the same with the reference's 45 epipolar bins, literally (matcher.h:218-331) */
uint32_t pgo_ref_guided_match_binned(const double F[9], const float* kp1, uint32_t n1, const float* kp2, uint32_t n2,
                                     const float* d1, const float* d2, uint32_t dim, const int size_src[2],
                                     const int size_dst[2], int n_bins, uint32_t* out_i, uint32_t* out_j, double* out_ratio,
                                     uint8_t* fragile);

/* batch over a flattened (pair,corr) SoA; OpenMP over pairs (threads<=0: all) */
void pgo_estimate_pose_batch(const float* x1, const float* y1, const float* x2,
                             const float* y2, const uint64_t* offsets, uint32_t n_pairs,
                             const double* thr, const double* guesses /*P*12 or NULL*/,
                             const uint8_t* has_guess, const pgo_params* prm, uint64_t seed,
                             uint64_t pair_id_base, pgo_edge* out, uint8_t* masks,
                             int threads);
int pgo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
