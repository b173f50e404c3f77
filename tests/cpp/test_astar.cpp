// Host-only test driver for host/graph_traversal.hpp: reads a pose graph + similarity matrix + queries,
// writes path, chained pose and touched-node count per query (compared with oracle/astar_oracle.py).
#include <chrono>
#include <cstdio>
#include <fstream>

#include "distributed.hpp"
#include "graph_traversal.hpp"
#include "tracklets.hpp"
#include "utils.hpp"

using namespace reconstruction;

// mode "formats": <similarity.txt> <N> <threshold> <list_with_focals.txt> <out.txt>
static int formats(char** argv) {
    const size_t N = (size_t)std::atoi(argv[3]);
    SimilarityTable sim(N, std::atof(argv[4]));
    std::FILE* out = std::fopen(argv[6], "w");
    std::fprintf(out, "load %d\n", (int)sim.loadFromFile(argv[2]));
    auto& q = sim.getMutablePrioritizedViewPairs();
    std::fprintf(out, "pairs %zu views %zu\n", q.size(), sim.getKeptViews().size());
    while (!q.empty()) {
        auto t = q.top();
        q.pop();
        std::fprintf(out, "%.3f %zu %zu\n", std::get<0>(t), (size_t)std::get<1>(t), (size_t)std::get<2>(t));
    }
    std::fprintf(out, "sim01 %.3f sim10 %.3f\n", sim.getSimilarity(0, 1), sim.getSimilarity(1, 0));
    size_t total = 0;
    std::vector<std::tuple<std::string, double, double, double>> imgs;
    std::fprintf(out, "list %d\n", (int)load1DSfMImageList(argv[5], total, imgs));
    std::fprintf(out, "total %zu\n", total);
    for (auto& t : imgs) std::fprintf(out, "%s %.4f\n", std::get<0>(t).c_str(), std::get<1>(t));
    RunningStatistics st;
    st.addTime("[Pose estimation]", 2.0);
    st.addTime("[Pose estimation]", 4.0);
    st.addCount("[Pose estimation] Runs", 1);
    st.addCount("[Pose estimation] Runs", 1);
    std::fprintf(out, "stat %.1f %zu %.1f\n", st.getTime("[Pose estimation]"), st.getCount("[Pose estimation] Runs"),
                 st.getAverageTime("[Pose estimation]").first);
    {   // the A* heuristic's per-search view of the table (costsTo(to)(next)) must be getCost(next, to) cell for cell --
        // on an asymmetric file that is the COLUMN of `to`, not its row (reference graph_traversal.h:847)
        ImageSimilarityHeuristics h(sim);
        const size_t N = (size_t)std::atoi(argv[3]);
        size_t bad = 0;
        for (size_t to = 0; to <= N; ++to) {
            const auto costs = h.costsTo(to);
            for (size_t next = 0; next <= N; ++next) bad += costs(next) != h.getCost(next, to);
        }
        std::fprintf(out, "costs_mismatch %zu c01 %.3f c10 %.3f\n", bad, h.costsTo(1)(0), h.costsTo(0)(1));
    }
    std::fclose(out);
    return 0;
}

// mode "tracklets": <commands.txt> <out.txt>; commands: "add src dst m  p1 p2 keep ..." | "get src dst max"
static int tracklets(char** argv) {
    std::FILE* in = std::fopen(argv[2], "r");
    std::FILE* out = std::fopen(argv[3], "w");
    if (!in || !out) return 3;
    Tracklets tr(64);
    char cmd[8];
    while (std::fscanf(in, "%7s", cmd) == 1) {
        size_t a, b, m;
        if (std::fscanf(in, "%zu %zu %zu", &a, &b, &m) != 3) return 4;
        if (cmd[0] == 'a') {
            std::vector<Tracklets::Match> matches(m);
            std::vector<unsigned char> mask(m);
            for (size_t k = 0; k < m; ++k) {
                size_t p1, p2;
                int keep;
                if (std::fscanf(in, "%zu %zu %d", &p1, &p2, &keep) != 3) return 4;
                matches[k] = Tracklets::Match(p1, p2, 0.0);
                mask[k] = (unsigned char)keep;
            }
            tr.add(a, b, matches, mask);
            std::fprintf(out, "tracks %zu\n", tr.trackNumber());
        } else {
            std::vector<Tracklets::Match> got;
            tr.getCorrespondences(got, a, b, m);
            std::fprintf(out, "get %zu", got.size());
            for (auto& g : got) std::fprintf(out, " %zu %zu", std::get<0>(g), std::get<1>(g));
            std::fprintf(out, "\n");
        }
    }
    std::fclose(in);
    std::fclose(out);
    return 0;
}

// mode "trackbench": <views> <points per view> <pairs>: nanoseconds per registered match of Tracklets::add on a
// consistent scene (keypoint p of every view sees 3-D point p): the host-side cost processFeatures pays per inlier
static int trackbench(char** argv) {
    const size_t V = (size_t)std::atoi(argv[2]), K = (size_t)std::atoi(argv[3]), pairs = (size_t)std::atoi(argv[4]);
    Tracklets tr(V);
    std::vector<Tracklets::Match> matches(K);
    std::vector<unsigned char> mask(K, 1);
    size_t total = 0, got = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t e = 0; e < pairs; ++e) {
        const size_t s = (e * 7) % V, d = (s + 1 + (e * 13) % (V - 1)) % V;
        for (size_t k = 0; k < K; ++k) matches[k] = Tracklets::Match((k * 31 + e) % K, (k * 31 + e) % K, 0.0);
        tr.add(s, d, matches, mask);
        total += K;
        std::vector<Tracklets::Match> q;
        tr.getCorrespondences(q, d, (s + 2) % V, 5000);
        got += q.size();
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("trackbench: %zu matches in %.3f s = %.1f ns per match; %zu tracks, %zu correspondences returned\n", total, sec,
                1e9 * sec / (double)total, tr.trackNumber(), got);
    return 0;
}

// mode "hostcomm": <out prefix>; RANK / WORLD_SIZE / MASTER_* from the environment.  Exercises the TCP star that
// bootstraps the multi-GPU path (host/distributed.hpp): broadcast, uneven all-gather-v, fixed-size all-gather, barrier.
static int hostcomm(char** argv) {
    const dist::LaunchEnv env = dist::LaunchEnv::fromEnvironment();
    dist::HostComm comm(env);
    uint8_t id[128];
    for (int i = 0; i < 128; ++i) id[i] = env.rank == 0 ? (uint8_t)(i * 7 + 3) : 0;
    comm.broadcast(id, sizeof id);
    std::vector<uint64_t> bytes(env.world);
    for (uint32_t r = 0; r < env.world; ++r) bytes[r] = r == 1 ? 0 : 1000 * (r + 1) + 13;  // one empty block
    std::vector<uint8_t> mine(bytes[env.rank]);
    for (size_t i = 0; i < mine.size(); ++i) mine[i] = (uint8_t)(env.rank * 31 + i);
    uint64_t total = 0;
    for (uint64_t b : bytes) total += b;
    std::vector<uint8_t> all(total);
    comm.allgatherv(mine.data(), mine.size(), all.data(), bytes.data());
    struct Rec { uint32_t rank; uint32_t sq; };
    const std::vector<Rec> recs = comm.allgather(Rec{env.rank, env.rank * env.rank});
    comm.barrier();
    std::ofstream out(std::string(argv[2]) + "." + std::to_string(env.rank), std::ios::binary);
    out.write((const char*)id, sizeof id);
    out.write((const char*)all.data(), (std::streamsize)all.size());
    out.write((const char*)recs.data(), (std::streamsize)(recs.size() * sizeof(Rec)));
    return 0;
}

// mode "shards": <sizes.txt> <world> <out.txt>: dist::shardBounds (must equal pyposegraphbuilder.distributed.shard_bounds)
static int shards(char** argv) {
    std::ifstream in(argv[2]);
    std::vector<uint64_t> sizes;
    uint64_t v;
    while (in >> v) sizes.push_back(v);
    std::FILE* out = std::fopen(argv[4], "w");
    for (auto& b : dist::shardBounds(sizes, (uint32_t)std::atoi(argv[3]))) std::fprintf(out, "%zu %zu\n", b.first, b.second);
    std::fclose(out);
    return 0;
}

// "graphops" mode of tests/cpp/test_astar.cpp: a script of PoseGraph operations in, their results out
static int graphops(char** argv) {
    std::ifstream in(argv[2]);
    std::ofstream out(argv[3]);
    PoseGraph g;
    std::string op;
    auto poseOf = [](double score) {
        SE3d T;
        T.t = {{score, 2.0 * score, -score}};
        return T;
    };
    while (in >> op) {
        if (op == "V") {
            ViewId v; in >> v;
            out << (g.addVertex(v) ? 1 : 0) << "\n";
        } else if (op == "P") {
            ViewId a, b; in >> a >> b;
            g.addVertexPair(a, b);
            out << g.numVertices() << "\n";
        } else if (op == "E") {
            ViewId s, d; double sc; in >> s >> d >> sc;
            out << (g.addEdge(s, d, Pose(poseOf(sc)), sc) ? 1 : 0) << "\n";
        } else if (op == "B") {
            size_t n; in >> n;
            std::vector<PoseGraph::NewEdge> items(n);
            std::vector<SE3d> poses(n);
            for (size_t i = 0; i < n; ++i) {
                in >> items[i].src >> items[i].dst >> items[i].score;
                poses[i] = poseOf(items[i].score);
            }
            for (size_t i = 0; i < n; ++i) { items[i].R = poses[i].R.data(); items[i].t = poses[i].t.data(); }
            out << g.addEdges(items.data(), n) << "\n";
        } else if (op == "T") {  // the same batch through the team overload (three parts, started here as threads)
            size_t n; in >> n;
            std::vector<PoseGraph::NewEdge> items(n);
            std::vector<SE3d> poses(n);
            for (size_t i = 0; i < n; ++i) {
                in >> items[i].src >> items[i].dst >> items[i].score;
                poses[i] = poseOf(items[i].score);
            }
            for (size_t i = 0; i < n; ++i) { items[i].R = poses[i].R.data(); items[i].t = poses[i].t.data(); }
            auto team = [](size_t parts, const std::function<void(size_t)>& fn) {
                std::vector<std::thread> th;
                for (size_t p = 1; p < parts; ++p) th.emplace_back(fn, p);
                fn(0);
                for (std::thread& t : th) t.join();
            };
            out << g.addEdges(items.data(), n, 3, team, 0) << "\n";
        } else if (op == "A") {  // admitPairs: n, then n x (src dst) -> the admit flags and the vertex count afterwards
            size_t n; in >> n;
            std::vector<ViewId> s(n), d(n);
            for (size_t i = 0; i < n; ++i) in >> s[i] >> d[i];
            std::vector<uint8_t> admit(n, 7);
            g.admitPairs(s.data(), d.data(), n, admit.data());
            for (size_t i = 0; i < n; ++i) out << (int)admit[i];
            out << " " << g.numVertices() << "\n";
        } else if (op == "Y") {  // anyEdgeBetween: n, then n x (src dst)
            size_t n; in >> n;
            std::vector<ViewId> s(n), d(n);
            for (size_t i = 0; i < n; ++i) in >> s[i] >> d[i];
            out << (g.anyEdgeBetween(s.data(), d.data(), n) ? 1 : 0) << "\n";
        } else if (op == "H") {
            ViewId s, d; in >> s >> d;
            out << (g.hasEdge(s, d) ? 1 : 0) << " " << (g.hasEdgeBetween(s, d) ? 1 : 0) << "\n";
        } else if (op == "G") {
            ViewId s, d; in >> s >> d;
            const PoseGraphEdge e = g.getEdgeById({s, d});
            if (e.isUndefined()) out << "none\n";
            else out << e.getSourceId() << " " << e.getDestinationId() << " " << e.getScore() << " " << e.getValue().getTranslation()[1] << "\n";
        } else if (op == "N") {
            ViewId v; in >> v;
            std::vector<EdgeId> ids;
            const bool has = g.getEdgesByVertex(v, ids);
            out << (has ? 1 : 0) << " " << g.getEdgeNumberByVertex(v);
            for (const EdgeId& id : ids) out << " " << id.first << ":" << id.second;
            out << "\n";
        } else if (op == "I") {
            out << g.numVertices() << " " << g.numEdges();
            for (const EdgeId& id : g.getEdgeIds()) out << " " << id.first << ":" << id.second;
            out << "\n";
        } else {
            return 3;
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 7 && std::string(argv[1]) == "formats") return formats(argv);
    if (argc >= 4 && std::string(argv[1]) == "graphops") return graphops(argv);
    if (argc >= 4 && std::string(argv[1]) == "tracklets") return tracklets(argv);
    if (argc >= 3 && std::string(argv[1]) == "hostcomm") {
        try {
            return hostcomm(argv);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "hostcomm: %s\n", e.what());
            return 1;
        }
    }
    if (argc >= 5 && std::string(argv[1]) == "shards") return shards(argv);
    if (argc >= 5 && std::string(argv[1]) == "trackbench") return trackbench(argv);
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
    std::vector<PoseGraph::NewEdge> batch;
    std::vector<Matrix3d> batch_R;
    std::vector<Vector3d> batch_t;
    for (uint32_t e = 0; e < E; ++e) {
        uint32_t s, d;
        SE3d T;
        double score;
        in.read((char*)&s, 4); in.read((char*)&d, 4);
        in.read((char*)T.R.data(), 72); in.read((char*)T.t.data(), 24); in.read((char*)&score, 8);
        g.addEdge(s, d, Pose(T), score);
        batch_R.push_back(T.R); batch_t.push_back(T.t);
        batch.push_back(PoseGraph::NewEdge{s, d, score, nullptr, nullptr});
    }
    {   // the same edges through addEdges (one locked batch per half, as the wave scheduler commits them) -- with an edge
        // given twice and one to a vertex the graph does not have, both of which must be refused like addEdge refuses them
        PoseGraph g2;
        for (uint32_t v = 0; v < V; ++v) g2.addVertex(v);
        for (size_t i = 0; i < batch.size(); ++i) { batch[i].R = batch_R[i].data(); batch[i].t = batch_t[i].data(); }
        std::vector<PoseGraph::NewEdge> first(batch.begin(), batch.begin() + batch.size() / 2), second(batch.begin() + batch.size() / 2, batch.end());
        if (!first.empty()) second.push_back(first[0]);
        if (!batch.empty()) second.push_back(PoseGraph::NewEdge{V + 7, 0, 0.5, batch_R[0].data(), batch_t[0].data()});
        const size_t added = g2.addEdges(first.data(), first.size()) + g2.addEdges(second.data(), second.size());
        bool same = added == g.numEdges() && g2.numEdges() == g.numEdges() && g2.getEdgeIds() == g.getEdgeIds();
        for (uint32_t v = 0; v < V && same; ++v) {
            std::vector<EdgeId> a, b;
            const bool ha = g.getEdgesByVertex(v, a), hb = g2.getEdgesByVertex(v, b);
            same = ha == hb && a == b;
        }
        for (const EdgeId& id : g.getEdgeIds()) {
            const PoseGraphEdge ea = g.getEdgeById(id), eb = g2.getEdgeById(id);
            same = same && ea.getScore() == eb.getScore() && ea.getValue().getRotation() == eb.getValue().getRotation() &&
                   ea.getValue().getTranslation() == eb.getValue().getTranslation() &&
                   ea.getValue().getEssentialMatrix() == eb.getValue().getEssentialMatrix();
        }
        if (!same) {
            std::fprintf(stderr, "addEdges differs from addEdge\n");
            return 5;
        }
    }
    ImageSimilarityHeuristics h(sim);
    AStarTraversal<ImageSimilarityHeuristics> astar(&g, h, weight, 0.0, depth);
    // the same searches over a SPARSE table holding the same values and without the graph's lock (what the wave scheduler
    // runs on kCoreNumber threads): paths, poses and touched-node counts must be identical
    SimilarityTable sparse(SimilarityTable::Sparse(), V, 0.0);
    for (uint32_t i = 0; i < V; ++i)
        for (uint32_t j = i + 1; j < V; ++j)
            if (sim.getSimilarity(i, j) != 0.0) sparse.setSimilarity(i, j, sim.getSimilarity(i, j));
    ImageSimilarityHeuristics hs(sparse);
    AStarTraversal<ImageSimilarityHeuristics> frozen(&g, hs, weight, 0.0, depth);
    frozen.setGraphFrozen(true);
    std::ofstream out(argv[2], std::ios::binary);
    for (uint32_t q = 0; q < Q; ++q) {
        uint32_t a, b;
        in.read((char*)&a, 4); in.read((char*)&b, 4);
        std::vector<ViewId> path;
        std::vector<SE3d> poses;
        size_t touched = 0, found = 0;
        bool exists = false;
        astar.getPath(a, b, path, poses, touched, found, exists);
        {
            std::vector<ViewId> path2;
            std::vector<SE3d> poses2;
            size_t touched2 = 0, found2 = 0;
            bool exists2 = false;
            frozen.getPath(a, b, path2, poses2, touched2, found2, exists2);
            if (path2 != path || touched2 != touched || found2 != found || exists2 != exists || poses2.size() != poses.size() ||
                (!poses.empty() && (poses2[0].R != poses[0].R || poses2[0].t != poses[0].t))) {
                std::fprintf(stderr, "query %u: the lock-free search over the sparse table differs\n", q);
                return 4;
            }
        }
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
