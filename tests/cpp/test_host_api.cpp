// C++ host-API test: reads a synthetic two-view batch written by tests/test_host_cpp.py, runs the
// reference-named interface (PoseGraphBuilder::estimatePose / estimatePoses / run, getInliers,
// InTraversalPoseTester::test, getPoseFromEssentialMatrix) on the GPU and writes results back.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <thread>

#include "pose_graph_builder.hpp"

using namespace reconstruction;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t P = 0;
    in.read((char*)&P, 4);
    std::vector<PoseGraphBuilder::ViewPair> pairs(P);
    std::vector<SE3d> gt(P);
    for (uint32_t i = 0; i < P; ++i) {
        uint32_t n;
        double thr;
        in.read((char*)&n, 4);
        in.read((char*)&thr, 8);
        in.read((char*)gt[i].R.data(), 72);
        in.read((char*)gt[i].t.data(), 24);
        pairs[i].src = i;
        pairs[i].dst = i + 1;
        pairs[i].similarity = 1.0 - 0.001 * i;
        pairs[i].normalizedThreshold = thr;
        pairs[i].correspondences = CorrespondenceMatrix((int)n);
        in.read((char*)pairs[i].correspondences.ptr(), (size_t)n * 32);
    }
    PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.5, 0.4, "images", "ws", "sim.txt", "focals.txt", true,
                             true, true);  // defaults of examples/cpp_example.cpp:32-66
    std::ofstream out(argv[2], std::ios::binary);
    // (1) the single-pair seam
    for (uint32_t i = 0; i < P; ++i) {
        SE3d T;
        std::vector<uchar> mask;
        size_t ninl = 0;
        const bool ok = builder.estimatePose(20, pairs[i].correspondences, pairs[i].normalizedThreshold, {}, T, mask, ninl,
                                             /*seed*/ 42, /*pairId*/ i);
        const uint32_t okv = ok, nv = (uint32_t)ninl, ms = (uint32_t)mask.size();
        out.write((const char*)&okv, 4);
        out.write((const char*)&nv, 4);
        out.write((const char*)T.R.data(), 72);
        out.write((const char*)T.t.data(), 24);
        out.write((const char*)&ms, 4);
        out.write((const char*)mask.data(), ms);
    }
    // (2) evaluator / tester / decomposition on pair 0 with the ground-truth pose
    {
        Engine& eng = builder.getEngine();
        EssentialMatrixEvaluator ev(eng);
        Pose p(gt[0]);
        std::vector<size_t> inl;
        ev.getInliers(pairs[0].correspondences, p.getEssentialMatrix(), 1.5 * pairs[0].normalizedThreshold, inl);
        InTraversalPoseTester<> tester(eng, 1.5 * pairs[0].normalizedThreshold, 5, &pairs[0].correspondences);
        size_t tn = 0;
        const bool tok = tester.test(gt[0], tn);
        Matrix3d R;
        Vector3d t;
        const int votes = pose::getPoseFromEssentialMatrix(eng, p.getEssentialMatrix(), pairs[0].correspondences, R, t);
        const uint32_t a = (uint32_t)inl.size(), b = tok, c = (uint32_t)tn, d = (uint32_t)votes;
        out.write((const char*)&a, 4); out.write((const char*)&b, 4); out.write((const char*)&c, 4); out.write((const char*)&d, 4);
        out.write((const char*)R.data(), 72);
        out.write((const char*)t.data(), 24);
    }
    // (3) the wave-scheduled run over all candidate pairs
    {
        PoseGraph graph;
        auto cand = pairs;
        builder.run(cand, graph, /*waveSize*/ 3);
        const uint32_t ne = (uint32_t)graph.numEdges(), nv = (uint32_t)graph.numVertices();
        out.write((const char*)&ne, 4);
        out.write((const char*)&nv, 4);
        for (auto& id : graph.getEdgeIds()) {
            const PoseGraphEdge e = graph.getEdgeById(id);
            const uint32_t s = (uint32_t)e.getSourceId(), dd = (uint32_t)e.getDestinationId();
            const double sc = e.getScore();
            out.write((const char*)&s, 4); out.write((const char*)&dd, 4); out.write((const char*)&sc, 8);
            out.write((const char*)e.getValue().getRotation().data(), 72);
        }
    }
    // (4) re-entrancy: processImages calls the seam from kCoreNumber OpenMP threads (pose_graph_builder.h:391-392).
    // Eight threads share one builder (one context: calls serialise), then two builders run side by side (two
    // contexts, two streams); every result must equal the single-threaded one.
    {
        std::vector<SE3d> ref(P);
        std::vector<size_t> refInl(P);
        std::vector<std::vector<uchar>> refMask(P);
        for (uint32_t i = 0; i < P; ++i)
            builder.estimatePose(20, pairs[i].correspondences, pairs[i].normalizedThreshold, {}, ref[i], refMask[i], refInl[i], 42, i);
        auto sweep = [&](PoseGraphBuilder& b, uint32_t first, uint32_t step, int* bad) {
            for (uint32_t i = first; i < P; i += step) {
                SE3d T;
                std::vector<uchar> mask;
                size_t ninl = 0;
                b.estimatePose(20, pairs[i].correspondences, pairs[i].normalizedThreshold, {}, T, mask, ninl, 42, i);
                if (ninl != refInl[i] || mask != refMask[i] || T.R != ref[i].R || T.t != ref[i].t) ++*bad;
            }
        };
        int bad[8] = {0};
        {
            std::vector<std::thread> th;
            for (uint32_t k = 0; k < 8; ++k) th.emplace_back(sweep, std::ref(builder), k, 8u, &bad[k]);
            for (auto& t : th) t.join();
        }
        uint32_t sharedBad = 0;
        for (int k = 0; k < 8; ++k) sharedBad += (uint32_t)bad[k];
        PoseGraphBuilder b2(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.5, 0.4, "", "", "", "", true, true, true);
        int badA = 0, badB = 0;
        std::thread ta(sweep, std::ref(builder), 0u, 1u, &badA), tb(sweep, std::ref(b2), 0u, 1u, &badB);
        ta.join();
        tb.join();
        const uint32_t twoBad = (uint32_t)(badA + badB);
        out.write((const char*)&sharedBad, 4);
        out.write((const char*)&twoBad, 4);
    }
    std::printf("host api ok: %u pairs\n", P);
    return 0;
}
