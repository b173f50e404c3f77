/* pgi_host.h -- C entry points of libpgi_host.so, the C++ host layer (host/pose_graph_builder.hpp: the reference's
 * PoseGraphBuilder with its A* edge scheduler, visibility table and pose graph over libpgi.so).
 *
 * The reference is a C++ class (src/pyposegraphbuilder/include/pose_graph_builder.h:25-171) whose Python package is
 * empty; these few functions let a non-C++ caller -- pyposegraphbuilder.PoseGraphBuilder.run -- drive the SAME
 * scheduler the C++ surface uses instead of re-implementing it: descending-similarity waves, A* pose guesses on the
 * graph committed by the previous waves (findPath, :785-862), batched guess screening and estimatePose, edge and
 * visibility update.  Plain pointers and sizes, caller-owned buffers, negative return = error (pgih_last_error). */
#ifndef PGI_HOST_H
#define PGI_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgih_builder pgih_builder;

/* the 17 constructor arguments of reconstruction::PoseGraphBuilder, in the reference's order
 * (pose_graph_builder.h:30-47; call site examples/cpp_example.cpp:86-103) */
typedef struct {
    uint64_t core_number, maximum_tracklet_number, maximum_search_depth, maximum_path_number, minimum_inlier_number,
        minimum_point_number, maximum_point_number_for_epipolar_hashing;
    double traversal_heuristics_weight, similarity_threshold, inlier_outlier_threshold;
    const char *image_path, *workspace_path, *similarity_graph_path, *focal_length_path;
    int32_t use_path_finding, use_gpu, use_epipolar_hashing;
} pgih_config;

typedef struct {           /* one edge of the resulting pose graph (pose_graph.h:28-60) */
    uint32_t src, dst;
    double score;          /* inliers / matches (pose_graph_builder.h:645-654) */
    double R[9], t[3];     /* T_dst_src, row-major rotation */
} pgih_graph_edge;

#define PGIH_STATS 16      /* pairs processed, edges added, A* searched / found / touched, poses from guesses, hypotheses,
                            * waves, graph edges, quirk-only guesses, rest 0 */

const char* pgih_last_error(void);
/* One process per GPU: restricts the calling thread -- and every thread it starts later -- to the CPUs of the NUMA node the
 * HIP device hangs on (device < 0: the current one), what `numactl --cpunodebind` does in a launcher; memory touched first
 * afterwards lands on that node.  Call it before the inputs are read.  Returns the node, or -1 when nothing was done (one node,
 * sysfs unreadable, refused by a cpuset, PGI_HOST_NUMA=0).  Never an error. */
int pgih_bind_process_to_device_node(int device);
pgih_builder* pgih_create(const pgih_config* cfg);   /* NULL on failure (no HIP device: there is no CPU fallback) */
void pgih_destroy(pgih_builder* b);
/* pgi_params.guess_mode of the builder's engine: 1 = rotation-guided re-estimation of chained poses (BASELINE config 5) */
int pgih_set_rotation_guided(pgih_builder* b, int on);
/* pgi_params.lo_graph_cut of the builder's engine: lambda * 64 > 0 (9 ~ the paper's 0.14) = graph-cut local optimisation, the "GC" of
 * GC-RANSAC (the refit's rows are the minimum cut of the spatial-coherence energy; include/pgi.h), 0 = off (the default) */
int pgih_set_graph_cut(pgih_builder* b, uint32_t lambda64);
/* progressive sampling (pgi_params.sampler = 1) in pgih_run_features over the matcher's ratio-sorted rows: on by default */
int pgih_set_progressive_sampling(pgih_builder* b, int on);
/* PoseGraphBuilder::run over caller-provided candidate pairs.  Pair p: views src[p] -> dst[p], retrieval similarity,
 * normalised threshold thr[p], correspondences rows [offsets[p], offsets[p+1]) of corr_aos (n x 4 doubles, the reference's
 * cv::Mat N x 4 CV_64F; read in place, never copied on the host).  n_views: every id must be below it (0 = derive it from
 * the ids, which must then stay below PGIH_MAX_VIEWS); offsets must not decrease and no pair may exceed INT_MAX rows --
 * violations are errors, nothing is allocated from unchecked caller data.  The similarity table A* uses holds the candidate
 * pairs' values (0 elsewhere; a sparse table).  seed: wave w of the run draws with seed + w (0 = what the C++ drivers use).
 * Writes at most edge_capacity edges (in insertion order), their number to *n_edges, PGIH_STATS counters to stats (may be NULL). */
#define PGIH_MAX_VIEWS (1u << 24)
int pgih_run_pairs(pgih_builder* b, uint32_t n_views, uint32_t n_pairs, const uint32_t* src, const uint32_t* dst,
                   const double* similarity, const double* thr, const uint64_t* offsets, const double* corr_aos,
                   uint32_t wave_size, uint64_t seed, pgih_graph_edge* edges, uint32_t edge_capacity, uint32_t* n_edges,
                   uint64_t* stats);

typedef struct {           /* one view of pgih_run_features (feature_utils.h:53-95: what keypoints.h5 / image_data.h5 hold) */
    const float* keypoints;    /* n x 2 pixel coordinates */
    const float* descriptors;  /* n x 128, row-major */
    uint32_t n, pad;
    double focal_length, width, height;
} pgih_view;

#define PGIH_FEATURE_STATS 24  /* the PGIH_STATS counters, then [16] descriptor-matching runs, [17] tracklet quick-matching runs,
                                * [18] guided-matching runs, [19] guided matches added, [20] tracks, [21] pairs with too few matches */
#define PGIH_STAGES 8          /* seconds: upload + prepare, quick matching, matching, correspondences, A*, pose estimation,
                                * guided matching, commit + tracklets */

/* PoseGraphBuilder::processFeatures -- the loop body of processImages (pose_graph_builder.h:391-709) on in-memory features:
 * per wave of candidate pairs (descending similarity) tracklet quick matching, descriptor matching, createCorrespondenceMatrix,
 * A* pose guesses, estimatePose, guided matching, tracklet / visibility / graph update, with the builder's 17 arguments.
 * device_tracklets != 0 keeps the tracklet store in HBM (the default of the C++ class).  Outputs as pgih_run_pairs;
 * stats (PGIH_FEATURE_STATS) and stage_seconds (PGIH_STAGES) may be NULL. */
int pgih_run_features(pgih_builder* b, uint32_t n_views, const pgih_view* views, uint32_t n_pairs, const uint32_t* src,
                      const uint32_t* dst, const double* similarity, uint32_t wave_size, int device_tracklets, pgih_graph_edge* edges,
                      uint32_t edge_capacity, uint32_t* n_edges, uint64_t* stats, double* stage_seconds);

#ifdef __cplusplus
}
#endif
#endif /* PGI_HOST_H */
