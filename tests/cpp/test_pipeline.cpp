// Feature-level pipeline on the GPU: reads views (keypoints, descriptors, camera), candidate pairs and a similarity
// matrix written by tests/test_feature_pipeline.py, runs PoseGraphBuilder::processFeatures in four configurations
// (plain; + path finding; + path finding + epipolar hashing/tracklets in HBM; the same with the host tracklet store) and
// writes statistics + edges.
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "graph_traversal.hpp"

using namespace reconstruction;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    PoseGraphBuilder::bindProcessToDeviceNode();  // (what a launcher's --cpunodebind does; before the features are read)
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t V, P, wave;
    in.read((char*)&V, 4); in.read((char*)&P, 4); in.read((char*)&wave, 4);
    SimilarityTable sim(V, 0.0, false);
    for (uint32_t i = 0; i < V; ++i)
        for (uint32_t j = 0; j < V; ++j) {
            double s;
            in.read((char*)&s, 8);
            if (i < j) sim.setSimilarity(i, j, s);
        }
    std::vector<PoseGraphBuilder::ViewFeatures> views(V);
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t n;
        in.read((char*)&n, 4);
        in.read((char*)&views[v].focalLength, 8); in.read((char*)&views[v].width, 8); in.read((char*)&views[v].height, 8);
        views[v].keypoints.resize((size_t)n * 2);
        views[v].descriptors.resize((size_t)n * 128);
        in.read((char*)views[v].keypoints.data(), (size_t)n * 8);
        in.read((char*)views[v].descriptors.data(), (size_t)n * 512);
    }
    std::vector<PoseGraphBuilder::CandidatePair> pairs(P);
    for (uint32_t i = 0; i < P; ++i) {
        uint32_t s, d;
        double simv;
        in.read((char*)&s, 4); in.read((char*)&d, 4); in.read((char*)&simv, 8);
        pairs[i] = PoseGraphBuilder::CandidatePair{s, d, simv};
    }
    if (!in) return 3;
    std::ofstream out(argv[2], std::ios::binary);
    // argv[3] (optional): the modes to run, e.g. "2" or "023" (default all four); PGI_DRIVER_REPS > 1 repeats every mode
    // inside this process -- one timing block per repetition, the last one warm -- and writes the last repetition's result
    const std::string modes = argc > 3 ? argv[3] : "0123";
    const int reps = std::max(1, std::atoi(std::getenv("PGI_DRIVER_REPS") ? std::getenv("PGI_DRIVER_REPS") : "1"));
    // mode 3 = mode 2 with the tracklets in the host store instead of HBM; mode 4 = mode 2 with rotation-guided re-estimation
    // of the chained poses (pgi_params.guess_mode = 1) instead of the reference's guess test
    for (int mode = 0; mode < 5; ++mode) {
        if (modes.find((char)('0' + mode)) == std::string::npos) continue;
        for (int rep = 0; rep < reps; ++rep) {
        const bool usePath = mode >= 1, useHashing = mode >= 2;
        // thresholds as in examples/cpp_example.cpp: 20 inliers, 50 points, 100 guided matches, 0.75 px
        PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", usePath, true, useHashing);
        builder.setDeviceTracklets(mode != 3);
        if (mode == 4) builder.setRotationGuidedGuesses(true);
        PoseGraph graph;
        auto cand = pairs;
        const auto t0 = std::chrono::steady_clock::now();
        const auto st = builder.processFeatures(views, cand, graph, wave, &sim);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("mode %d: %zu pairs -> %zu edges in %.3f s (%.1f pairs/s; %zu matched, %zu quick, %zu guided runs)\n", mode,
                    (size_t)st.pairsProcessed, (size_t)st.edgesAdded, sec, st.pairsProcessed / sec, (size_t)st.matchingRuns,
                    (size_t)st.quickMatchingRuns, (size_t)st.guidedMatchingRuns);
        std::printf("        seconds: upload + prepare %.3f, quick matching %.3f, matching %.3f, correspondences %.3f, A* %.3f, pose estimation %.3f, guided %.3f, "
                    "commit + tracklets %.3f\n", st.secUpload, st.secQuickMatching, st.secMatching, st.secCorrespondences, st.secAStar,
                    st.secPoseEstimation, st.secGuidedMatching, st.secTrackUpdate);
        if (rep + 1 < reps) continue;
        const uint64_t v[16] = {st.pairsProcessed, st.edgesAdded, st.pathsSearched, st.pathsFound, st.touchedNodes, st.posesFromGuess,
                                st.hypotheses, st.waves, graph.numEdges(), st.matchingRuns, st.quickMatchingRuns, st.guidedMatchingRuns,
                                st.guidedMatchesAdded, st.trackNumber, st.tooFewMatches,
                                builder.getStatistics().getCount("[Pose estimation] Quirk-only guesses")};
        out.write((const char*)v, sizeof v);
        for (auto& id : graph.getEdgeIds()) {
            const PoseGraphEdge e = graph.getEdgeById(id);
            const uint32_t s = (uint32_t)id.first, d = (uint32_t)id.second;
            const double sc = e.getScore();
            out.write((const char*)&s, 4); out.write((const char*)&d, 4); out.write((const char*)&sc, 8);
            out.write((const char*)e.getValue().getRotation().data(), 72);
            out.write((const char*)e.getValue().getTranslation().data(), 24);
        }
        }
    }
    return 0;
}
