// hdf5_cache.hpp -- the reference's on-disk caches read and written with the HDF5 C library.
//
// The reference goes through cv::hdf (OpenCV contrib, absent here); the files themselves are plain HDF5: every
// cv::Mat is a 2-D dataset [rows, cols] of the native element type.
//   image_data.h5       "<image file name>"      1 x 3 f64  [focal, width, height]   pose_graph_builder.h:298-350
//   keypoints.h5        "feat_<name>"            K x 4 f64  [x, y, angle, size]      feature_utils.h:27-52, 81-95
//                       "desc_<name>"            K x 128 f32 RootSIFT
//                       attribute "finished"     int 1                                 pose_graph_builder.h:1100-1168
//   correspondences.h5  "<src name>_<dst name>"  M x 3 f64  [queryIdx, trainIdx, ratio] feature_utils.h:113-133, 190-206
// runWorkspace() is the reference's PoseGraphBuilder::run (pose_graph_builder.h:173-239): image list + focal lengths,
// image sizes, similarity matrix, features -> processFeatures.  Optional component: built only where hdf5.h exists.
#pragma once
#include <hdf5.h>

#include <fstream>
#include <limits>

#include "graph_traversal.hpp"
#include "reconstruction.hpp"
#include "utils.hpp"

namespace reconstruction {
namespace cache {

class Hdf5File {
   public:
    // cv::hdf::open semantics: open read-write, create when missing
    explicit Hdf5File(const std::string& path, bool readOnly = false) {
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);  // failures are reported through return values
        std::ifstream probe(path);
        const bool there = probe.is_open();
        probe.close();
        if (there) file = H5Fopen(path.c_str(), readOnly ? H5F_ACC_RDONLY : H5F_ACC_RDWR, H5P_DEFAULT);
        else if (!readOnly) file = H5Fcreate(path.c_str(), H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
    }
    ~Hdf5File() { close(); }
    Hdf5File(const Hdf5File&) = delete;
    Hdf5File& operator=(const Hdf5File&) = delete;
    bool isOpen() const { return file >= 0; }
    void close() {
        if (file >= 0) H5Fclose(file);
        file = -1;
    }
    bool hlexists(const std::string& name) const { return file >= 0 && H5Lexists(file, name.c_str(), H5P_DEFAULT) > 0; }
    bool atexists(const std::string& name) const { return file >= 0 && H5Aexists(file, name.c_str()) > 0; }
    bool atwrite(int value, const std::string& name) {
        if (file < 0) return false;
        const hid_t sp = H5Screate(H5S_SCALAR);
        const hid_t at = H5Acreate2(file, name.c_str(), H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT);
        const bool ok = at >= 0 && H5Awrite(at, H5T_NATIVE_INT, &value) >= 0;
        if (at >= 0) H5Aclose(at);
        H5Sclose(sp);
        return ok;
    }
    // dsread: any numeric 2-D (or 1-D) dataset, converted by the library to T (double or float)
    template <typename T>
    bool dsread(const std::string& name, std::vector<T>& data, size_t& rows, size_t& cols) const {
        if (!hlexists(name)) return false;
        const hid_t ds = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
        if (ds < 0) return false;
        const hid_t sp = H5Dget_space(ds);
        hsize_t dims[3] = {1, 1, 1};
        const int nd = H5Sget_simple_extent_ndims(sp);
        bool ok = nd >= 1 && nd <= 3 && H5Sget_simple_extent_dims(sp, dims, nullptr) >= 0;
        if (ok) {
            rows = (size_t)dims[0];
            cols = nd >= 2 ? (size_t)(dims[1] * (nd == 3 ? dims[2] : 1)) : 1;
            data.resize(rows * cols);
            ok = data.empty() || H5Dread(ds, memType<T>(), H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data()) >= 0;
        }
        H5Sclose(sp);
        H5Dclose(ds);
        return ok;
    }
    // dscreate + dswrite of a rows x cols single-channel matrix
    template <typename T>
    bool dswrite(const std::string& name, const T* data, size_t rows, size_t cols) {
        if (file < 0 || hlexists(name)) return false;
        const hsize_t dims[2] = {rows, cols};
        const hid_t sp = H5Screate_simple(2, dims, nullptr);
        const hid_t ds = H5Dcreate2(file, name.c_str(), memType<T>(), sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        const bool ok = ds >= 0 && (rows * cols == 0 || H5Dwrite(ds, memType<T>(), H5S_ALL, H5S_ALL, H5P_DEFAULT, data) >= 0);
        if (ds >= 0) H5Dclose(ds);
        H5Sclose(sp);
        return ok;
    }

   private:
    template <typename T>
    static hid_t memType();
    hid_t file = -1;
};
template <>
inline hid_t Hdf5File::memType<double>() { return H5T_NATIVE_DOUBLE; }
template <>
inline hid_t Hdf5File::memType<float>() { return H5T_NATIVE_FLOAT; }

typedef std::tuple<std::string, double, double, double> ImageData;  // (file name, focal, width, height)

// pose_graph_builder.h:298-350: image sizes come from image_data.h5 when the image is listed there, otherwise the
// record is written (the reference gets the size from cv::imread; here the caller must have filled it in).
inline bool loadImageData(const std::string& kImageDatabaseFilename_, std::vector<ImageData>& imageData_) {
    Hdf5File db(kImageDatabaseFilename_);
    if (!db.isOpen()) return false;
    for (ImageData& d : imageData_) {
        const std::string& name = std::get<0>(d);
        if (name.empty()) continue;
        std::vector<double> v;
        size_t r = 0, c = 0;
        if (db.hlexists(name)) {
            if (!db.dsread(name, v, r, c) || v.size() < 3) return false;
            std::get<2>(d) = v[1];
            std::get<3>(d) = v[2];
        } else {
            const double rec[3] = {std::get<1>(d), std::get<2>(d), std::get<3>(d)};
            if (!db.dswrite(name, rec, 1, 3)) return false;
        }
    }
    return true;
}

// feature_utils.h:27-52 (load) / 81-95 (save); names are the image names without extension
inline bool loadFeatures(const Hdf5File& db, const std::string& imageName_, PoseGraphBuilder::ViewFeatures& f) {
    std::vector<double> kp;
    size_t r = 0, c = 0, dr = 0, dc = 0;
    if (!db.dsread("feat_" + imageName_, kp, r, c) || c < 2) return false;
    if (!db.dsread("desc_" + imageName_, f.descriptors, dr, dc) || dr != r || (r && dc != PGI_DESC_DIM)) return false;
    f.keypoints.resize(2 * r);
    for (size_t i = 0; i < r; ++i) {  // cv::KeyPoint::pt is float
        f.keypoints[2 * i] = (float)kp[i * c];
        f.keypoints[2 * i + 1] = (float)kp[i * c + 1];
    }
    return true;
}
inline bool saveFeatures(Hdf5File& db, const std::string& imageName_, const PoseGraphBuilder::ViewFeatures& f) {
    const size_t n = f.size();
    std::vector<double> kp(4 * n, 0.0);  // angle and size are not used by this path
    for (size_t i = 0; i < n; ++i) { kp[4 * i] = f.keypoints[2 * i]; kp[4 * i + 1] = f.keypoints[2 * i + 1]; }
    return db.dswrite("feat_" + imageName_, kp.data(), n, 4) && db.dswrite("desc_" + imageName_, f.descriptors.data(), n, PGI_DESC_DIM);
}

// feature_utils.h:113-133 (load) / 190-206 (save)
inline bool loadCorrespondences(const Hdf5File& db, const std::string& src, const std::string& dst,
                                std::vector<std::tuple<size_t, size_t, double>>& matches_) {
    std::vector<double> m;
    size_t r = 0, c = 0;
    if (!db.dsread(src + "_" + dst, m, r, c) || c < 3) return false;
    matches_.reserve(r);
    for (size_t i = 0; i < r; ++i) matches_.emplace_back((size_t)m[i * c], (size_t)m[i * c + 1], m[i * c + 2]);
    return true;
}
inline bool saveCorrespondences(Hdf5File& db, const std::string& src, const std::string& dst,
                                const std::vector<std::tuple<size_t, size_t, double>>& matches_) {
    std::vector<double> m(3 * matches_.size());
    for (size_t i = 0; i < matches_.size(); ++i) {
        m[3 * i] = (double)std::get<0>(matches_[i]);
        m[3 * i + 1] = (double)std::get<1>(matches_[i]);
        m[3 * i + 2] = std::get<2>(matches_[i]);
    }
    return db.dswrite(src + "_" + dst, m.data(), matches_.size(), 3);
}

// PoseGraphBuilder::run (pose_graph_builder.h:173-239) on a workspace holding image_data.h5, keypoints.h5 and,
// optionally, correspondences.h5.  Keypoint DETECTION is out of scope: keypoints.h5 must be finished (:1100-1168).
inline PoseGraphBuilder::FeatureRunStatistics runWorkspace(PoseGraphBuilder& builder, const std::string& kFocalLengthPath,
                                                           const std::string& kSimilarityGraphPath, const std::string& kWorkspacePath,
                                                           double kSimilarityThreshold, PoseGraph& poseGraph_, size_t waveSize = 1024,
                                                           Reconstruction* reconstruction_ = nullptr) {
    std::vector<ImageData> imageData;
    size_t totalImageNumber = 0;
    if (!load1DSfMImageList(kFocalLengthPath, totalImageNumber, imageData)) throw PgiError("cannot read " + kFocalLengthPath);
    if (!loadImageData(kWorkspacePath + "image_data.h5", imageData)) throw PgiError("cannot read image_data.h5");
    if (reconstruction_) builder.initializeReconstruction(imageData.size(), imageData, *reconstruction_, poseGraph_);  // :213-217
    Hdf5File keypointDb(kWorkspacePath + "keypoints.h5", /*readOnly*/ true);
    if (!keypointDb.isOpen() || !keypointDb.atexists("finished")) throw PgiError("keypoints.h5 missing or not finished");
    const size_t V = imageData.size();
    std::vector<PoseGraphBuilder::ViewFeatures> views(V);
    std::vector<std::string> names(V);
    for (size_t v = 0; v < V; ++v) {
        const std::string& file = std::get<0>(imageData[v]);
        names[v] = file.size() > 4 ? file.substr(0, file.size() - 4) : file;  // :258-260: cut the extension
        views[v].focalLength = std::get<1>(imageData[v]);
        views[v].width = std::get<2>(imageData[v]);
        views[v].height = std::get<3>(imageData[v]);
        poseGraph_.addVertex(v);  // :292
        if (views[v].focalLength <= std::numeric_limits<double>::epsilon()) continue;  // :1138-1139: unknown focal length
        if (!loadFeatures(keypointDb, names[v], views[v])) views[v] = PoseGraphBuilder::ViewFeatures();
    }
    SimilarityTable similarityTable(totalImageNumber, kSimilarityThreshold);
    if (!similarityTable.loadFromFile(kSimilarityGraphPath)) throw PgiError("cannot read " + kSimilarityGraphPath);
    std::vector<PoseGraphBuilder::CandidatePair> pairs;
    auto& heap = similarityTable.getMutablePrioritizedViewPairs();
    while (!heap.empty()) {
        const auto t = heap.top();
        heap.pop();
        const ViewId a = std::get<1>(t), b = std::get<2>(t);
        if (a >= V || b >= V || views[a].size() == 0 || views[b].size() == 0) continue;
        pairs.push_back(PoseGraphBuilder::CandidatePair{a, b, std::get<0>(t)});
    }
    Hdf5File correspondenceDb(kWorkspacePath + "correspondences.h5", /*readOnly*/ true);
    const PoseGraphBuilder::MatchLookup lookup = [&](ViewId s, ViewId d, std::vector<std::tuple<size_t, size_t, double>>& m) {
        return correspondenceDb.isOpen() && loadCorrespondences(correspondenceDb, names[s], names[d], m);
    };
    return builder.processFeatures(views, pairs, poseGraph_, waveSize, &similarityTable, correspondenceDb.isOpen() ? &lookup : nullptr);
}

}  // namespace cache

// The reference's outer entry point (pose_graph_builder.h:69-71, 173-239) with its own argument list: everything
// else comes from the 17 constructor arguments.  Ends like the reference (:711-714): statistics, then the edge count.
inline void PoseGraphBuilder::run(Reconstruction& reconstruction_, PoseGraph& poseGraph_) {
    cache::runWorkspace(*this, kFocalLengthPath, kSimilarityGraphPath, kWorkspacePath, kSimilarityThreshold, poseGraph_, 1024,
                        &reconstruction_);
    statistics.print();
    std::printf("Edges in the pose-graph = %d\n", (int)poseGraph_.numEdges());
}

}  // namespace reconstruction
