/*
 * pgi_reference_adapter.h -- the binding a maintainer of danini/pose-graph-initialization adds.
 *
 * Header-only C++17, compiled INSIDE the reference tree, where OpenCV, Eigen and Sophus exist.  It maps the
 * reference's own types onto the C ABI of include/pgi.h so that the bodies of
 *     PoseGraphBuilder::estimatePose                      src/pyposegraphbuilder/include/pose_graph_builder.h:940-1078
 *     EssentialMatrixEvaluator::getInliers                 .../graph_traversal.h:136-168
 *     InTraversalPoseTester::test                          .../graph_traversal.h:194-233
 *     pose::getPoseFromEssentialMatrix                     .../pose_utils.h:172-252
 * become one call each; signatures, ownership (caller-owned outputs resized by the callee) and the bool error
 * convention are the reference's.  cv::Mat N x 4 CV_64F is exactly `corr_aos`; Sophus::SE3d is (R row-major, t);
 * std::vector<uchar> is the mask buffer.
 *
 * On this image none of the three libraries is installed, so the header compiles to nothing here (guarded by
 * __has_include); tests/test_host_cpp.py checks that it at least preprocesses, and the same mapping is exercised for
 * real through the dependency-free host layer (pose-graph-initialization_amd/host/pose_graph_builder.hpp).
 */
#ifndef PGI_REFERENCE_ADAPTER_H
#define PGI_REFERENCE_ADAPTER_H

#include "pgi.h"

#if defined(__cplusplus) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && __has_include(<sophus/se3.hpp>) && __has_include(<Eigen/Core>)
#define PGI_REFERENCE_ADAPTER_AVAILABLE 1

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core.hpp>
#include <sophus/se3.hpp>

#include <cstdint>
#include <stdexcept>
#include <vector>

namespace reconstruction {
namespace mi355x {

/* One context per process.  pgi_estimate_pose is re-entrant (a pool of private stream + staging slots), so the
 * kCoreNumber OpenMP threads of processImages (pose_graph_builder.h:391-392) share it and overlap on the GPU. */
inline pgi_ctx* context() {
    static pgi_ctx* ctx = [] {
        pgi_ctx* c = pgi_create(/*device*/ -1, /*params: defaults*/ nullptr);
        if (!c) throw std::runtime_error(pgi_last_error());  /* no HIP device: there is no CPU fallback */
        return c;
    }();
    return ctx;
}

typedef Eigen::Matrix<double, 3, 3, Eigen::RowMajor> RowMajor3d;

/* Replaces the body of PoseGraphBuilder::estimatePose (pose_graph_builder.h:940-1078).  kReconstruction_, the pixel
 * threshold and the view indices are unused by the reference's body; seed / pairId select the RNG stream (use the
 * pair's position in the similarity queue for run-to-run reproducibility). */
inline bool estimatePose(const size_t kMinimumInlierNumber_, const cv::Mat& kCorrespondences_ /* N x 4 CV_64F */,
                         const double kThreshold_ /* normalised, :934-937 */, const std::vector<Sophus::SE3d>& poseGuesses_,
                         Sophus::SE3d& estimatedPose_, std::vector<uchar>& inlierMask_, size_t& inlierNumber_,
                         const uint64_t seed = 0, const uint64_t pairId = 0) {
    CV_Assert(kCorrespondences_.type() == CV_64F && kCorrespondences_.cols == 4 && kCorrespondences_.isContinuous());
    std::vector<double> g(12 * poseGuesses_.size());
    for (size_t i = 0; i < poseGuesses_.size(); ++i) {  /* row-major R, then t */
        Eigen::Map<RowMajor3d>(&g[12 * i]) = poseGuesses_[i].rotationMatrix();
        Eigen::Map<Eigen::Vector3d>(&g[12 * i + 9]) = poseGuesses_[i].translation();
    }
    inlierMask_.assign((size_t)kCorrespondences_.rows, 0);  /* :1000, :1034 */
    pgi_edge e;
    uchar none = 0;
    const int rc = pgi_estimate_pose(context(), kCorrespondences_.ptr<double>(), (uint32_t)kCorrespondences_.rows, kThreshold_,
                                     g.empty() ? nullptr : g.data(), (uint32_t)poseGuesses_.size(),
                                     (uint32_t)kMinimumInlierNumber_, seed, pairId, &e,
                                     inlierMask_.empty() ? &none : inlierMask_.data());
    if (rc < 0) throw std::runtime_error(pgi_last_error());
    inlierNumber_ = e.n_inl;                 /* :1022, :1047 */
    if (rc != 1) return false;               /* :1053-1054 (too few inliers), :1069-1070 (NaN) */
    const Eigen::Matrix3d R = Eigen::Map<const RowMajor3d>(e.R);
    estimatedPose_ = Sophus::SE3d(Eigen::Quaterniond(R), Eigen::Map<const Eigen::Vector3d>(e.t));  /* :1073-1075 */
    return true;
}

/* EssentialMatrixEvaluator::getInliers (graph_traversal.h:136-168).  The reference compares the SQUARED Sampson
 * distance with the un-squared kThreshold_ (line 164); pass kThreshold_ through unchanged to reproduce that.
 * One re-entrant call (pgi_score_pose_f64_host leases a private slot: no hipMalloc, no shared stream). */
inline void getInliers(const cv::Mat& kCorrespondences_, const Eigen::Matrix3d& kDescriptor_, const double& kThreshold_,
                       std::vector<size_t>& inliers_) {
    CV_Assert(kCorrespondences_.type() == CV_64F && kCorrespondences_.cols == 4 && kCorrespondences_.isContinuous());
    const RowMajor3d E = kDescriptor_;
    std::vector<uchar> mask((size_t)kCorrespondences_.rows, 0);
    uint32_t count = 0;
    uchar none = 0;
    const int rc = pgi_score_pose_f64_host(context(), kCorrespondences_.ptr<double>(), (uint32_t)kCorrespondences_.rows, E.data(),
                                           kThreshold_, 0, &count, mask.empty() ? &none : mask.data());
    if (rc < 0) throw std::runtime_error(pgi_last_error());
    inliers_.reserve(count);
    for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i]) inliers_.emplace_back(i);
}

/* InTraversalPoseTester::test (graph_traversal.h:194-233): true at kMinimumInlierNumber_ inliers of the squared bound;
 * the scan stops at that inlier (:221-225), on the device too. */
inline bool testPose(const cv::Mat& kCorrespondences_, const Eigen::Matrix3d& kEssential_, const double kSquaredThreshold_,
                     const size_t kMinimumInlierNumber_, size_t& inlierNumber_) {
    CV_Assert(kCorrespondences_.type() == CV_64F && kCorrespondences_.cols == 4 && kCorrespondences_.isContinuous());
    const RowMajor3d E = kEssential_;
    uint32_t count = 0;
    const int rc = pgi_score_pose_f64_host(context(), kCorrespondences_.ptr<double>(), (uint32_t)kCorrespondences_.rows, E.data(),
                                           kSquaredThreshold_, (uint32_t)kMinimumInlierNumber_, &count, nullptr);
    if (rc < 0) throw std::runtime_error(pgi_last_error());
    inlierNumber_ = count;
    return rc == 1;
}

/* pose::getPoseFromEssentialMatrix (pose_utils.h:172-252): E and the N x 4 CV_64F correspondence matrix in; rotation, unit
 * translation out; returns the winning candidate's vote count (:251).  Every row votes, as in the reference (:203).  One
 * re-entrant call on host pointers (pgi_pose_from_essential_host).  Candidate order, first maximum and the translation
 * sign rule are the reference's; the per-row test is the product's depth-sign rule. */
inline int getPoseFromEssentialMatrix(const Eigen::Matrix3d& essential_matrix_, const cv::Mat& normalized_correspondences_,
                                      Eigen::Matrix3d& rotation_, Eigen::Vector3d& translation_) {
    CV_Assert(normalized_correspondences_.type() == CV_64F && normalized_correspondences_.cols == 4 &&
              normalized_correspondences_.isContinuous());
    const RowMajor3d E = essential_matrix_;
    double R[9], t[3];
    uint32_t votes = 0;
    const int rc = pgi_pose_from_essential_host(context(), E.data(),
                                                normalized_correspondences_.rows ? normalized_correspondences_.ptr<double>() : nullptr,
                                                (uint32_t)normalized_correspondences_.rows, /*all rows vote*/ nullptr, R, t, &votes, nullptr);
    if (rc < 0) throw std::runtime_error(pgi_last_error());
    rotation_ = Eigen::Map<const RowMajor3d>(R);
    translation_ = Eigen::Map<const Eigen::Vector3d>(t);
    return (int)votes;
}

}  // namespace mi355x
}  // namespace reconstruction

#endif /* __has_include(...) */
#endif /* __cplusplus && __has_include */

#ifndef PGI_REFERENCE_ADAPTER_AVAILABLE
#define PGI_REFERENCE_ADAPTER_AVAILABLE 0 /* OpenCV / Eigen / Sophus headers not found: nothing is declared */
#endif

#endif /* PGI_REFERENCE_ADAPTER_H */
