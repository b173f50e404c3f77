// c_api.cpp -- the C entry points of libpgi_host.so (include/pgi_host.h): PoseGraphBuilder::run for non-C++ callers.
#include "../../include/pgi_host.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <memory>
#include <string>

#include "graph_traversal.hpp"

using namespace reconstruction;

struct pgih_builder {
    std::unique_ptr<PoseGraphBuilder> impl;
};

namespace {
thread_local std::string g_error;
int fail(const std::string& what) {
    g_error = what;
    return -1;
}
// No C++ exception may cross the C ABI: every entry point runs its body through one of these (anything that is not a
// std::exception -- a foreign exception type, a thrown integer -- is reported too instead of unwinding into the caller).
template <class Body>
int guarded(const char* who, Body body) {
    try {
        return body();
    } catch (const std::exception& e) {
        return fail(std::string(who) + ": " + e.what());
    } catch (...) {
        return fail(std::string(who) + ": unknown exception");
    }
}

void exportGraph(const PoseGraph& graph, pgih_graph_edge* edges, uint32_t edge_capacity, uint32_t* n_edges) {
    uint32_t k = 0;
    for (const EdgeId& id : graph.getEdgeIds()) {
        if (k < edge_capacity) {
            const PoseGraphEdge e = graph.getEdgeById(id);
            pgih_graph_edge& o = edges[k];
            o.src = (uint32_t)id.first;
            o.dst = (uint32_t)id.second;
            o.score = e.getScore();
            for (int c = 0; c < 9; ++c) o.R[c] = e.getValue().getRotation()[c];
            for (int c = 0; c < 3; ++c) o.t[c] = e.getValue().getTranslation()[c];
        }
        ++k;
    }
    *n_edges = k;
}
}  // namespace

extern "C" {

const char* pgih_last_error(void) { return g_error.c_str(); }

int pgih_bind_process_to_device_node(int device) {
    int node = -1;
    guarded("pgih_bind_process_to_device_node", [&]() { node = reconstruction::PoseGraphBuilder::bindProcessToDeviceNode(device); return 0; });
    return node;
}

pgih_builder* pgih_create(const pgih_config* c) {
    if (!c) {
        g_error = "pgih_create: null configuration";
        return nullptr;
    }
    pgih_builder* made = nullptr;
    guarded("pgih_create", [&]() {
        auto str = [](const char* s) { return std::string(s ? s : ""); };
        std::unique_ptr<pgih_builder> b(new pgih_builder);
        b->impl.reset(new PoseGraphBuilder(c->core_number, c->maximum_tracklet_number, c->maximum_search_depth, c->maximum_path_number,
                                           c->minimum_inlier_number, c->minimum_point_number, c->maximum_point_number_for_epipolar_hashing,
                                           c->traversal_heuristics_weight, c->similarity_threshold, c->inlier_outlier_threshold,
                                           str(c->image_path), str(c->workspace_path), str(c->similarity_graph_path),
                                           str(c->focal_length_path), c->use_path_finding != 0, c->use_gpu != 0,
                                           c->use_epipolar_hashing != 0));
        made = b.release();
        return 0;
    });
    return made;
}

void pgih_destroy(pgih_builder* b) {
    guarded("pgih_destroy", [&]() {
        delete b;
        return 0;
    });
}

int pgih_set_rotation_guided(pgih_builder* b, int on) {
    if (!b) return fail("pgih_set_rotation_guided: null builder");
    return guarded("pgih_set_rotation_guided", [&]() {
        b->impl->setRotationGuidedGuesses(on != 0);
        return 0;
    });
}

int pgih_set_graph_cut(pgih_builder* b, uint32_t lambda64) {
    if (!b) return fail("pgih_set_graph_cut: null builder");
    return guarded("pgih_set_graph_cut", [&]() {
        b->impl->setGraphCutLocalOptimisation(lambda64);
        return 0;
    });
}

int pgih_set_progressive_sampling(pgih_builder* b, int on) {
    if (!b) return fail("pgih_set_progressive_sampling: null builder");
    return guarded("pgih_set_progressive_sampling", [&]() {
        b->impl->setProgressiveSampling(on != 0);
        return 0;
    });
}

int pgih_run_pairs(pgih_builder* b, uint32_t n_views, uint32_t n_pairs, const uint32_t* src, const uint32_t* dst,
                   const double* similarity, const double* thr, const uint64_t* offsets, const double* corr_aos, uint32_t wave_size,
                   uint64_t seed, pgih_graph_edge* edges, uint32_t edge_capacity, uint32_t* n_edges, uint64_t* stats) {
    if (!b || !n_edges || (n_pairs && (!src || !dst || !similarity || !thr || !offsets)) || (edge_capacity && !edges))
        return fail("pgih_run_pairs: null argument");
    return guarded("pgih_run_pairs", [&]() {
        // the caller's ids size host tables (visibility, A* stamps): they are checked, never trusted
        uint32_t views = n_views;
        if (!views) {
            for (uint32_t p = 0; p < n_pairs; ++p) {
                const uint32_t m = std::max(src[p], dst[p]);
                if (m >= PGIH_MAX_VIEWS) return fail("pgih_run_pairs: view id " + std::to_string(m) + " with n_views = 0 (derive) exceeds PGIH_MAX_VIEWS");
                views = std::max(views, m + 1u);
            }
        } else if (views > PGIH_MAX_VIEWS) {
            return fail("pgih_run_pairs: n_views exceeds PGIH_MAX_VIEWS");
        }
        for (uint32_t p = 0; p < n_pairs; ++p) {
            if (src[p] >= views || dst[p] >= views) return fail("pgih_run_pairs: pair " + std::to_string(p) + ": view index out of range");
            if (offsets[p + 1] < offsets[p]) return fail("pgih_run_pairs: offsets must not decrease (pair " + std::to_string(p) + ")");
            if (offsets[p + 1] - offsets[p] > (uint64_t)INT32_MAX) return fail("pgih_run_pairs: pair " + std::to_string(p) + " has more than INT_MAX rows");
        }
        if (n_pairs && offsets[n_pairs] > offsets[0] && !corr_aos) return fail("pgih_run_pairs: null correspondences");
        // the table A* reads: the candidate pairs' values, 0 elsewhere -- a sparse table, not V x V doubles
        SimilarityTable sim(SimilarityTable::Sparse(), std::max(views, 1u), 0.0);
        std::vector<PoseGraphBuilder::ViewPair> pairs(n_pairs);
        for (uint32_t p = 0; p < n_pairs; ++p) {
            PoseGraphBuilder::ViewPair& vp = pairs[p];
            vp.src = src[p];
            vp.dst = dst[p];
            vp.similarity = similarity[p];
            vp.normalizedThreshold = thr[p];
            // a header over the caller's rows (cv::Mat over foreign memory): the batch is not copied on the host
            vp.correspondences = CorrespondenceMatrix::viewOf(corr_aos + 4 * offsets[p], (int)(offsets[p + 1] - offsets[p]));
            sim.setSimilarity(vp.src, vp.dst, vp.similarity);
        }
        PoseGraph graph;
        const PoseGraphBuilder::RunStatistics st = b->impl->run(pairs, graph, wave_size ? wave_size : 4096, &sim, seed);
        exportGraph(graph, edges, edge_capacity, n_edges);
        if (stats) {
            const uint64_t v[PGIH_STATS] = {st.pairsProcessed, st.edgesAdded, st.pathsSearched, st.pathsFound, st.touchedNodes,
                                            st.posesFromGuess, st.hypotheses, st.waves, graph.numEdges(), st.quirkOnlyGuesses};
            std::copy(v, v + PGIH_STATS, stats);
        }
        if (*n_edges > edge_capacity) return fail("pgih_run_pairs: edge buffer too small");
        return 0;
    });
}

int pgih_run_features(pgih_builder* b, uint32_t n_views, const pgih_view* views, uint32_t n_pairs, const uint32_t* src,
                      const uint32_t* dst, const double* similarity, uint32_t wave_size, int device_tracklets, pgih_graph_edge* edges,
                      uint32_t edge_capacity, uint32_t* n_edges, uint64_t* stats, double* stage_seconds) {
    if (!b || !n_edges || (n_views && !views) || (n_pairs && (!src || !dst || !similarity)) || (edge_capacity && !edges))
        return fail("pgih_run_features: null argument");
    return guarded("pgih_run_features", [&]() {
        std::vector<PoseGraphBuilder::ViewFeaturesRef> vf(n_views);
        for (uint32_t v = 0; v < n_views; ++v) {
            vf[v].keypoints = views[v].keypoints;
            vf[v].descriptors = views[v].descriptors;
            vf[v].n = views[v].n;
            vf[v].focalLength = views[v].focal_length; vf[v].width = views[v].width; vf[v].height = views[v].height;
        }
        SimilarityTable sim(SimilarityTable::Sparse(), std::max(n_views, 1u), 0.0);
        std::vector<PoseGraphBuilder::CandidatePair> cand(n_pairs);
        for (uint32_t p = 0; p < n_pairs; ++p) {
            if (src[p] >= n_views || dst[p] >= n_views) return fail("pgih_run_features: view index out of range");
            cand[p] = PoseGraphBuilder::CandidatePair{src[p], dst[p], similarity[p]};
            sim.setSimilarity(src[p], dst[p], similarity[p]);
        }
        b->impl->setDeviceTracklets(device_tracklets != 0);
        PoseGraph graph;
        const PoseGraphBuilder::FeatureRunStatistics st = b->impl->processFeatures(vf, cand, graph, wave_size ? wave_size : 512, &sim);
        exportGraph(graph, edges, edge_capacity, n_edges);
        if (stats) {
            const uint64_t v[PGIH_FEATURE_STATS] = {st.pairsProcessed, st.edgesAdded, st.pathsSearched, st.pathsFound, st.touchedNodes,
                                                    st.posesFromGuess, st.hypotheses, st.waves, graph.numEdges(),
                                                    b->impl->getStatistics().getCount("[Pose estimation] Quirk-only guesses"), 0, 0, 0, 0, 0, 0,
                                                    st.matchingRuns, st.quickMatchingRuns, st.guidedMatchingRuns, st.guidedMatchesAdded,
                                                    st.trackNumber, st.tooFewMatches};
            std::copy(v, v + PGIH_FEATURE_STATS, stats);
        }
        if (stage_seconds) {
            const double sec[PGIH_STAGES] = {st.secUpload, st.secQuickMatching, st.secMatching, st.secCorrespondences, st.secAStar,
                                             st.secPoseEstimation, st.secGuidedMatching, st.secTrackUpdate};
            std::copy(sec, sec + PGIH_STAGES, stage_seconds);
        }
        if (*n_edges > edge_capacity) return fail("pgih_run_features: edge buffer too small");
        return 0;
    });
}

}  // extern "C"
