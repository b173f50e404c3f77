// HDF5 caches + workspace runner (host/hdf5_cache.hpp).
//   roundtrip <dir>            no GPU: write and re-read image_data / keypoints / correspondences datasets
//   workspace <scene.bin> <dir> <out.txt>   GPU: write the scene of tests/test_feature_pipeline.py as a 1DSfM-style workspace
//                              (list_with_focals.txt, similarity matrix text, *.h5), run it through runWorkspace and
//                              compare with processFeatures on the in-memory features
#include <cstdio>
#include <cstring>
#include <fstream>

#include "hdf5_cache.hpp"

using namespace reconstruction;
typedef PoseGraphBuilder::ViewFeatures Features;

static int roundtrip(const std::string& dir) {
    Features f;
    for (int i = 0; i < 37; ++i) {
        f.keypoints.push_back(0.25f * i); f.keypoints.push_back(100.5f - i);
        for (int k = 0; k < 128; ++k) f.descriptors.push_back((float)((i * 131 + k * 7) % 97) / 97.0f);
    }
    const std::string kp = dir + "/keypoints.h5", im = dir + "/image_data.h5", co = dir + "/correspondences.h5";
    {
        cache::Hdf5File db(kp);
        if (!db.isOpen() || !cache::saveFeatures(db, "img_a", f) || db.atexists("finished") || !db.atwrite(1, "finished")) return 10;
        if (cache::saveFeatures(db, "img_a", f)) return 11;  // datasets are created once (dscreate fails on an existing name)
    }
    {
        cache::Hdf5File db(kp, true);
        Features g;
        if (!db.atexists("finished") || !db.hlexists("feat_img_a") || db.hlexists("feat_img_b")) return 12;
        if (!cache::loadFeatures(db, "img_a", g) || g.keypoints != f.keypoints || g.descriptors != f.descriptors) return 13;
        if (cache::loadFeatures(db, "img_b", g)) return 14;
    }
    std::vector<cache::ImageData> data = {cache::ImageData("img_a.jpg", 1234.5, 1600.0, 1200.0), cache::ImageData("img_b.jpg", 900.0, 800.0, 600.0)};
    if (!cache::loadImageData(im, data)) return 15;  // first call writes the records
    std::vector<cache::ImageData> again = {cache::ImageData("img_a.jpg", 1234.5, 0.0, 0.0), cache::ImageData("img_b.jpg", 900.0, 0.0, 0.0)};
    if (!cache::loadImageData(im, again) || std::get<2>(again[0]) != 1600.0 || std::get<3>(again[1]) != 600.0) return 16;
    std::vector<std::tuple<size_t, size_t, double>> m = {{3, 9, 0.5}, {7, 1, 0.625}}, r;
    {
        cache::Hdf5File db(co);
        if (!cache::saveCorrespondences(db, "img_a", "img_b", m)) return 17;
    }
    cache::Hdf5File db(co, true);
    if (!cache::loadCorrespondences(db, "img_a", "img_b", r) || r != m || cache::loadCorrespondences(db, "img_b", "img_a", r)) return 18;
    std::printf("roundtrip ok\n");
    return 0;
}

static void edgesOf(const PoseGraph& g, std::vector<double>& out) {
    for (auto& id : g.getEdgeIds()) {
        const PoseGraphEdge e = g.getEdgeById(id);
        out.push_back((double)id.first); out.push_back((double)id.second); out.push_back(e.getScore());
        for (int c = 0; c < 9; ++c) out.push_back(e.getValue().getRotation()[c]);
        for (int c = 0; c < 3; ++c) out.push_back(e.getValue().getTranslation()[c]);
    }
}

static int workspace(const char* scene, const std::string& dir, const char* outPath) {
    std::ifstream in(scene, std::ios::binary);
    uint32_t V, P, wave;
    in.read((char*)&V, 4); in.read((char*)&P, 4); in.read((char*)&wave, 4);
    std::vector<double> sim((size_t)V * V);
    in.read((char*)sim.data(), sim.size() * 8);
    std::vector<Features> views(V);
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t n;
        in.read((char*)&n, 4);
        in.read((char*)&views[v].focalLength, 8); in.read((char*)&views[v].width, 8); in.read((char*)&views[v].height, 8);
        views[v].keypoints.resize((size_t)n * 2);
        views[v].descriptors.resize((size_t)n * 128);
        in.read((char*)views[v].keypoints.data(), (size_t)n * 8);
        in.read((char*)views[v].descriptors.data(), (size_t)n * 512);
    }
    if (!in) return 3;
    // the workspace a 1DSfM run leaves behind
    std::vector<cache::ImageData> data;
    {
        std::ofstream list(dir + "/list_with_focals.txt");
        std::ofstream simf(dir + "/similarity.txt");
        cache::Hdf5File kp(dir + "/keypoints.h5");
        for (uint32_t v = 0; v < V; ++v) {
            char name[32];
            std::snprintf(name, sizeof name, "view%03u", v);
            list << "images/" << name << ".jpg 0 " << views[v].focalLength << "\n";
            data.emplace_back(std::string(name) + ".jpg", views[v].focalLength, views[v].width, views[v].height);
            if (!cache::saveFeatures(kp, name, views[v])) return 4;
            for (uint32_t j = 0; j < V; ++j) {
                char buf[16];
                std::snprintf(buf, sizeof buf, "%1.3f", v == j ? 1.0 : sim[(size_t)v * V + j]);
                simf << buf << (j + 1 < V ? " " : "\n");
            }
        }
        if (!kp.atwrite(1, "finished") || !cache::loadImageData(dir + "/image_data.h5", data)) return 5;
    }
    std::FILE* out = std::fopen(outPath, "w");
    // (a) in-memory reference: the candidates the similarity file yields, through processFeatures
    std::vector<double> ea, eb, ec;
    PoseGraphBuilder::FeatureRunStatistics sa, sb, sc;
    ViewId top_a = 0, top_b = 1;  // the most similar pair: first wave, never tracklet-matched
    {
        PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", true, true, true);
        SimilarityTable table(V, 0.05);
        if (!table.loadFromFile(dir + "/similarity.txt")) return 6;
        std::vector<PoseGraphBuilder::CandidatePair> pairs;
        auto& heap = table.getMutablePrioritizedViewPairs();
        while (!heap.empty()) { pairs.push_back({std::get<1>(heap.top()), std::get<2>(heap.top()), std::get<0>(heap.top())}); heap.pop(); }
        top_a = pairs[0].src; top_b = pairs[0].dst;
        PoseGraph g;
        for (uint32_t v = 0; v < V; ++v) g.addVertex(v);
        sa = builder.processFeatures(views, pairs, g, wave, &table);
        edgesOf(g, ea);
    }
    // (b) the same run from the files
    {
        PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", true, true, true);
        PoseGraph g;
        sb = cache::runWorkspace(builder, dir + "/list_with_focals.txt", dir + "/similarity.txt", dir + "/", 0.05, g, wave);
        edgesOf(g, eb);
    }
    // (d) the reference's own call shape (examples/cpp_example.cpp:82-106), unchanged apart from the includes:
    //     17 constructor arguments, then builder.run(reconstruction, poseGraph)
    std::vector<double> ed;
    int shape_ok = 0;
    {
        reconstruction::Reconstruction reconstruction;
        reconstruction::PoseGraph poseGraph;

        reconstruction::PoseGraphBuilder builder(
            20,      // FLAGS_core_number
            5000,    // FLAGS_maximum_tracklet_number
            5,       // FLAGS_maximum_search_depth
            100,     // FLAGS_maximum_path_number
            20,      // FLAGS_minimum_inlier_number
            50,      // FLAGS_minimum_point_number
            100,     // FLAGS_maximum_points_from_epipolar_hashing
            0.8,     // FLAGS_traversal_heuristics_weight
            0.05,    // FLAGS_similarity_threshold
            0.75,    // FLAGS_inlier_outlier_threshold
            dir + "/images/",               // FLAGS_image_path
            dir + "/",                      // FLAGS_workspace_path
            dir + "/similarity.txt",        // FLAGS_similarity_graph_path
            dir + "/list_with_focals.txt",  // FLAGS_focal_length_path
            true,    // FLAGS_use_path_finding
            true,    // FLAGS_use_gpu
            true);   // FLAGS_use_epipolar_hashing

        builder.run(reconstruction,
            poseGraph);

        edgesOf(poseGraph, ed);
        // statistics to compare with: the same workspace through runWorkspace at run()'s own wave size
        PoseGraphBuilder::FeatureRunStatistics sd;
        {
            PoseGraphBuilder b2(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", true, true, true);
            PoseGraph g2;
            sd = cache::runWorkspace(b2, dir + "/list_with_focals.txt", dir + "/similarity.txt", dir + "/", 0.05, g2, 1024);
            std::vector<double> e2;
            edgesOf(g2, e2);
            if (e2 != ed) std::fprintf(stderr, "reference call shape: check edges-vs-runWorkspace failed\n");
        }
        // what initializeReconstruction leaves behind (pose_graph_builder.h:241-291) and the observability keys (:505-699)
        const PinholeCamera cam0 = reconstruction.getCamera(0);
        const View view0 = reconstruction.getView(0);
        RunningStatistics& rs = builder.getStatistics();
        const bool checks[] = {
            reconstruction.getViewNumber() == V, reconstruction.getCameraIds().size() == V, poseGraph.numVertices() == V,
            cam0.getIntrinsics()[0] == views[0].focalLength, cam0.getIntrinsics()[2] == views[0].width / 2.0,
            cam0.getIntrinsics()[5] == views[0].height / 2.0, cam0.getWidth() == views[0].width,
            view0.getMetadata().count("name") && view0.getMetadata().at("name") == "view000",
            view0.getMetadata().count("extension") && view0.getMetadata().at("extension") == "jpg",
            reconstruction.getView(V + 7).id() == UndefinedViewParameter,
            rs.getCount("[Matching] Runs") == sd.matchingRuns, rs.getCount("[Quick matching] Runs") == sd.quickMatchingRuns,
            rs.getCount("[Pose estimation] Runs") > 0, rs.getCount("[Pose estimation] Inlier number") > 0,
            rs.getCount("[A*] Runs") == sd.pathsSearched, rs.getCount("[A*] Touched nodes") == sd.touchedNodes,
            rs.getCount("[Epipolar Hashing] Runs") == sd.guidedMatchingRuns,
            rs.getCount("[Epipolar Hashing] Correspondences added") == sd.guidedMatchesAdded,
            rs.getTime("[Pose estimation]") > 0.0, rs.getCount("[Visibility update] Runs") == sd.pairsProcessed};
        shape_ok = 1;
        for (size_t k = 0; k < sizeof checks / sizeof checks[0]; ++k)
            if (!checks[k]) {
                shape_ok = 0;
                std::fprintf(stderr, "reference call shape: check %zu failed\n", k);
            }
    }
    // (c) with a correspondences.h5 holding a deliberately tiny match list for the most similar pair
    {
        cache::Hdf5File co(dir + "/correspondences.h5");
        std::vector<std::tuple<size_t, size_t, double>> few = {{0, 0, 0.1}, {1, 1, 0.2}};
        char na[32], nb[32];
        std::snprintf(na, sizeof na, "view%03u", (unsigned)top_a);
        std::snprintf(nb, sizeof nb, "view%03u", (unsigned)top_b);
        if (!cache::saveCorrespondences(co, na, nb, few)) return 7;
    }
    {
        PoseGraphBuilder builder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", true, true, true);
        PoseGraph g;
        sc = cache::runWorkspace(builder, dir + "/list_with_focals.txt", dir + "/similarity.txt", dir + "/", 0.05, g, wave);
        edgesOf(g, ec);
        std::fprintf(out, "edge_with_cached_tiny_list %d\n", (int)(g.hasEdge(top_a, top_b) || g.hasEdge(top_b, top_a)));
    }
    std::fprintf(out, "edges_a %zu edges_b %zu identical %d\n", ea.size() / 15, eb.size() / 15, (int)(ea == eb));
    std::fprintf(out, "reference_call_shape edges %zu reconstruction_and_statistics %d\n", ed.size() / 15, shape_ok);
    std::fprintf(out, "stats_a %zu %zu %zu %zu\n", sa.pairsProcessed, sa.edgesAdded, sa.matchingRuns, sa.quickMatchingRuns);
    std::fprintf(out, "stats_b %zu %zu %zu %zu\n", sb.pairsProcessed, sb.edgesAdded, sb.matchingRuns, sb.quickMatchingRuns);
    std::fprintf(out, "stats_c %zu %zu cached %zu toofew %zu\n", sc.pairsProcessed, sc.edgesAdded, sc.cachedMatchLoads, sc.tooFewMatches);
    std::fclose(out);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "roundtrip")) return roundtrip(argv[2]);
    if (argc >= 5 && !std::strcmp(argv[1], "workspace")) return workspace(argv[2], argv[3], argv[4]);
    return 2;
}
