// Wave scheduler + A* pose guesses on the GPU (BASELINE config 5 surrogate): reads a scene graph written by
// tests/test_scheduler.py, runs PoseGraphBuilder::run without and with path finding, writes statistics + edges.
#include <cstdio>
#include <fstream>

#include "graph_traversal.hpp"

using namespace reconstruction;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
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
    std::vector<PoseGraphBuilder::ViewPair> pairs(P);
    for (uint32_t i = 0; i < P; ++i) {
        uint32_t s, d, n;
        double thr, simv;
        in.read((char*)&s, 4); in.read((char*)&d, 4); in.read((char*)&n, 4); in.read((char*)&thr, 8); in.read((char*)&simv, 8);
        pairs[i].src = s; pairs[i].dst = d; pairs[i].similarity = simv; pairs[i].normalizedThreshold = thr;
        pairs[i].correspondences = CorrespondenceMatrix((int)n);
        in.read((char*)pairs[i].correspondences.ptr(), (size_t)n * 32);
    }
    std::ofstream out(argv[2], std::ios::binary);
    for (int usePath = 0; usePath < 2; ++usePath) {
        PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.4, "", "", "", "", usePath != 0, true, true);
        PoseGraph graph;
        auto cand = pairs;
        const auto st = builder.run(cand, graph, wave, &sim);
        const uint64_t v[9] = {st.pairsProcessed, st.edgesAdded, st.pathsSearched, st.pathsFound, st.touchedNodes,
                               st.posesFromGuess, st.hypotheses, st.waves, graph.numEdges()};
        out.write((const char*)v, sizeof v);
        for (auto& id : graph.getEdgeIds()) {
            const PoseGraphEdge e = graph.getEdgeById(id);
            const uint32_t s = (uint32_t)id.first, d = (uint32_t)id.second;
            const double sc = e.getScore();
            out.write((const char*)&s, 4); out.write((const char*)&d, 4); out.write((const char*)&sc, 8);
            out.write((const char*)e.getValue().getRotation().data(), 72);
        }
    }
    return 0;
}
