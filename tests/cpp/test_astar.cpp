// Host-only test driver for host/graph_traversal.hpp: reads a pose graph + similarity matrix + queries,
// writes path, chained pose and touched-node count per query (compared with oracle/astar_oracle.py).
#include <cstdio>
#include <fstream>

#include "graph_traversal.hpp"

using namespace reconstruction;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t V, E, Q;
    double weight;
    uint32_t depth;
    in.read((char*)&V, 4); in.read((char*)&E, 4); in.read((char*)&Q, 4); in.read((char*)&weight, 8); in.read((char*)&depth, 4);
    PoseGraph g;
    for (uint32_t v = 0; v < V; ++v) g.addVertex(v);
    SimilarityTable sim(V, 0.0, false);
    for (uint32_t i = 0; i < V; ++i)
        for (uint32_t j = 0; j < V; ++j) {
            double s;
            in.read((char*)&s, 8);
            if (i < j) sim.setSimilarity(i, j, s);
        }
    for (uint32_t e = 0; e < E; ++e) {
        uint32_t s, d;
        SE3d T;
        double score;
        in.read((char*)&s, 4); in.read((char*)&d, 4);
        in.read((char*)T.R.data(), 72); in.read((char*)T.t.data(), 24); in.read((char*)&score, 8);
        g.addEdge(s, d, Pose(T), score);
    }
    ImageSimilarityHeuristics h(sim);
    AStarTraversal<ImageSimilarityHeuristics> astar(&g, h, weight, 0.0, depth);
    std::ofstream out(argv[2], std::ios::binary);
    for (uint32_t q = 0; q < Q; ++q) {
        uint32_t a, b;
        in.read((char*)&a, 4); in.read((char*)&b, 4);
        std::vector<ViewId> path;
        std::vector<SE3d> poses;
        size_t touched = 0, found = 0;
        bool exists = false;
        astar.getPath(a, b, path, poses, touched, found, exists);
        const uint32_t n = exists ? (uint32_t)path.size() : 0, t = (uint32_t)touched;
        out.write((const char*)&n, 4);
        out.write((const char*)&t, 4);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t v = (uint32_t)path[i];
            out.write((const char*)&v, 4);
        }
        if (exists) {
            out.write((const char*)poses[0].R.data(), 72);
            out.write((const char*)poses[0].t.data(), 24);
        }
    }
    return 0;
}
