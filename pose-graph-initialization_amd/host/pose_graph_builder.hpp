// pose_graph_builder.hpp -- C++17 host layer over the C ABI (include/pgi.h), mirroring the
// reference's interface for the pose-estimation path so that its call sites read the same:
//
//   reconstruction::Pose                     <- include/pose.h:11-104
//   reconstruction::PoseGraphEdge/PoseGraph  <- include/pose_graph.h:28-226
//   reconstruction::EssentialMatrixEvaluator <- include/graph_traversal.h:82-170
//   reconstruction::InTraversalPoseTester    <- include/graph_traversal.h:174-242
//   reconstruction::pose::getPoseFromEssentialMatrix <- include/pose_utils.h:172-252
//   reconstruction::PoseGraphBuilder         <- include/pose_graph_builder.h:25-171 (17-argument ctor,
//                                               estimatePose seam :153-164, batched variant for the scheduler)
//
// The reference's signatures use cv::Mat / Eigen / Sophus types, none of which exist on this
// image; the stand-ins below keep the same member names and meaning:
//   Sophus::SE3d  -> reconstruction::SE3d  {rotationMatrix(), translation(), inverse(), operator*}
//   cv::Mat N x 4 CV_64F -> reconstruction::CorrespondenceMatrix (row-major doubles, rows/cols/ptr)
//   Eigen::Matrix3d -> reconstruction::Matrix3d (row-major double[9])
// All arithmetic runs on the GPU through libpgi.so; there is no CPU path.
#pragma once
#include <array>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/pgi.h"
#include "utils.hpp"

namespace reconstruction {

class Reconstruction;  // host/reconstruction.hpp
namespace dist {
class HostComm;  // host/distributed.hpp
}

typedef size_t ViewId;  // include/types.h:9-14
typedef std::pair<ViewId, ViewId> EdgeId;
typedef unsigned char uchar;
using Matrix3d = std::array<double, 9>;  // row-major
using Vector3d = std::array<double, 3>;

struct SE3d {  // T_dst_src: x_dst = R x_src + t
    Matrix3d R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    Vector3d t{{0, 0, 0}};
    SE3d() = default;
    SE3d(const Matrix3d& R_, const Vector3d& t_) : R(R_), t(t_) {}
    const Matrix3d& rotationMatrix() const { return R; }
    const Vector3d& translation() const { return t; }
    SE3d inverse() const {
        SE3d o;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o.R[3 * i + j] = R[3 * j + i];
        for (int i = 0; i < 3; ++i) o.t[i] = -(o.R[3 * i] * t[0] + o.R[3 * i + 1] * t[1] + o.R[3 * i + 2] * t[2]);
        return o;
    }
    SE3d operator*(const SE3d& b) const {  // (this * b)(x) = this(b(x))
        SE3d o;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                o.R[3 * i + j] = R[3 * i] * b.R[j] + R[3 * i + 1] * b.R[3 + j] + R[3 * i + 2] * b.R[6 + j];
            o.t[i] = R[3 * i] * b.t[0] + R[3 * i + 1] * b.t[1] + R[3 * i + 2] * b.t[2] + t[i];
        }
        return o;
    }
};

struct CorrespondenceMatrix {  // cv::Mat N x 4 CV_64F: [x1 y1 x2 y2], normalised (pose_graph_builder.h:917-931)
    int rows = 0;
    static constexpr int cols = 4;
    std::vector<double> data;
    const double* external = nullptr;  // rows owned by the caller (a cv::Mat header over foreign memory): see viewOf
    CorrespondenceMatrix() = default;
    explicit CorrespondenceMatrix(int n) : rows(n), data((size_t)n * 4) {}
    // a header over n rows the caller keeps alive (what cv::Mat(rows, 4, CV_64F, ptr) is): nothing is copied
    static CorrespondenceMatrix viewOf(const double* p, int n) {
        CorrespondenceMatrix m;
        m.rows = n;
        m.external = p;
        return m;
    }
    double* ptr(int r = 0) { return data.data() + 4 * (size_t)r; }  // owning matrices only
    const double* ptr(int r = 0) const { return (external ? external : data.data()) + 4 * (size_t)r; }
};

namespace pose {
// pose_utils.h:74-86
inline Matrix3d getEssentialMatrixFromRelativePose(const SE3d& p) {
    const Vector3d& t = p.t;
    const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    Matrix3d E;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += tx[3 * i + k] * p.R[3 * k + j];
            E[3 * i + j] = s;
        }
    return E;
}
}  // namespace pose

class Pose {  // include/pose.h
   public:
    Pose() = default;
    Pose(const SE3d& T_dst_src_, double source_scale_ = 0.0, double destination_scale_ = 0.0)
        : essentialMatrix(pose::getEssentialMatrixFromRelativePose(T_dst_src_)),
          T_dst_src(T_dst_src_),
          sourceScale(source_scale_),
          destinationScale(destination_scale_) {}
    Pose(const Matrix3d& rotation_, const Vector3d& translation_) : Pose(SE3d(rotation_, translation_)) {}
    Pose clone() const { return Pose(T_dst_src, sourceScale, destinationScale); }
    void setPose(const SE3d& T) {
        T_dst_src = T;
        essentialMatrix = pose::getEssentialMatrixFromRelativePose(T);
    }
    const SE3d& getTransform() const { return T_dst_src; }
    const Matrix3d& getRotation() const { return T_dst_src.R; }
    const Vector3d& getTranslation() const { return T_dst_src.t; }
    const Matrix3d& getEssentialMatrix() const { return essentialMatrix; }
    Pose getInverse() const { return Pose(T_dst_src.inverse()); }
    Pose operator*(const Pose& o) const { return Pose(T_dst_src * o.getTransform()); }
    void getScales(double& s, double& d) const {
        s = sourceScale;
        d = destinationScale;
    }

   protected:
    Matrix3d essentialMatrix{};
    SE3d T_dst_src;
    double sourceScale = 0, destinationScale = 0;
};

class PoseGraphVertex {
   public:
    static constexpr ViewId Undefined = (ViewId)-1;
    PoseGraphVertex(ViewId id_ = Undefined) : view_id(id_) {}
    const ViewId& id() const { return view_id; }
    bool isUndefined() const { return view_id == Undefined; }

   protected:
    ViewId view_id;
};

class PoseGraphEdge {  // include/pose_graph.h:28-60; score = inlier ratio (pose_graph_builder.h:645-654)
   public:
    PoseGraphEdge(ViewId s = PoseGraphVertex::Undefined, ViewId d = PoseGraphVertex::Undefined, Pose T = Pose(),
                  double score_ = 1.0)
        : view_id_src(s), view_id_dst(d), score(score_), T_dst_src(std::move(T)) {}
    bool isUndefined() const {
        return view_id_src == PoseGraphVertex::Undefined || view_id_dst == PoseGraphVertex::Undefined;
    }
    const ViewId& getSourceId() const { return view_id_src; }
    const ViewId& getDestinationId() const { return view_id_dst; }
    const Pose& getValue() const { return T_dst_src; }
    Pose& getMutableValue() { return T_dst_src; }
    const double& getScore() const { return score; }

   protected:
    ViewId view_id_src, view_id_dst;
    double score;
    Pose T_dst_src;
};

// The edge records of a PoseGraph: one array in insertion order that grows WITHOUT constructing its new tail -- addEdges' team
// writes the records of a wave concurrently (round 5), each thread touching its part of the fresh pages first.
class EdgeArray {
   public:
    EdgeArray() = default;
    EdgeArray(const EdgeArray&) = delete;
    EdgeArray& operator=(const EdgeArray&) = delete;
    ~EdgeArray() { std::free(p_); }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    void reserve(size_t cap) {
        static_assert(std::is_trivially_copyable<PoseGraphEdge>::value && std::is_trivially_destructible<PoseGraphEdge>::value,
                      "the records move with memcpy and are never destroyed one by one");
        if (cap <= cap_) return;
        PoseGraphEdge* q = static_cast<PoseGraphEdge*>(std::malloc(cap * sizeof(PoseGraphEdge)));
        if (!q) throw std::bad_alloc();
        if (n_) std::memcpy(static_cast<void*>(q), static_cast<const void*>(p_), n_ * sizeof(PoseGraphEdge));
        std::free(p_);
        p_ = q;
        cap_ = cap;
    }
    PoseGraphEdge& operator[](size_t k) { return p_[k]; }
    const PoseGraphEdge& operator[](size_t k) const { return p_[k]; }
    template <class... A>
    void emplace_back(A&&... a) {
        if (n_ == cap_) reserve(std::max<size_t>(16, 2 * cap_));
        new (p_ + n_++) PoseGraphEdge(std::forward<A>(a)...);
    }
    // m more records whose storage the caller constructs (every one of them, before anyone reads); returns the first
    PoseGraphEdge* extend(size_t m) {
        if (n_ + m > cap_) reserve(std::max(n_ + m, 2 * cap_));
        PoseGraphEdge* first = p_ + n_;
        n_ += m;
        return first;
    }

   private:
    PoseGraphEdge* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

class PoseGraph {  // include/pose_graph.h:62-226 (value semantics for missing ids)
   public:
    // Storage (round 4): the edges live in ONE array in insertion order; the id -> edge and vertex -> edges relations are
    // hash tables of indices into it.  The reference keeps three std::maps keyed by ids (pose_graph.h:214-224), which costs
    // the A* traversal a tree descent per visited edge; at SURVEY 8d's density (10^5 edges, ~40 per view) that was the
    // scheduler's largest host cost.  Interface and iteration orders (insertion order everywhere) are unchanged.
    size_t numVertices() const {
        std::shared_lock<std::shared_mutex> l(mu);
        return vertex_count;
    }
    size_t numEdges() const {
        std::shared_lock<std::shared_mutex> l(mu);
        return edge_store.size();
    }
    bool addVertex(ViewId id) {
        std::unique_lock<std::shared_mutex> l(mu);
        if (hasVertexUnlocked(id)) return false;
        if (id < kDenseIds) {
            if (id >= vertex_dense.size()) vertex_dense.resize(std::max<size_t>(id + 1, 2 * vertex_dense.size()), 0);
            vertex_dense[id] = 1;
        } else {
            vertex_sparse.emplace(id, PoseGraphVertex(id));
        }
        ++vertex_count;
        return true;
    }
    bool hasVertex(ViewId id) const {
        std::shared_lock<std::shared_mutex> l(mu);
        return hasVertexUnlocked(id);
    }
    PoseGraphVertex getVertexById(ViewId id) const {
        std::shared_lock<std::shared_mutex> l(mu);
        return hasVertexUnlocked(id) ? PoseGraphVertex(id) : PoseGraphVertex();
    }
    bool hasEdge(ViewId s, ViewId d) const {
        std::shared_lock<std::shared_mutex> l(mu);
        return edge_index.count({s, d}) != 0;
    }
    // (s, d) or (d, s): the scheduler's "is this pair already an edge?" (pose_graph_builder.h:426-431) under one lock
    bool hasEdgeBetween(ViewId s, ViewId d) const {
        std::shared_lock<std::shared_mutex> l(mu);
        return edge_index.count({s, d}) != 0 || edge_index.count({d, s}) != 0;
    }
    // Wave formation's two graph steps for MANY candidates under ONE lock (pose_graph_builder.h:426-431 "is this pair an edge
    // already, in either direction?" and, if not, pose_graph.h's addVertex for both ends): admit[i] = 1 and both vertices
    // added when pair i is not yet an edge, 0 (and nothing added) otherwise.  At 10^5 candidates the two locked calls per
    // candidate were 0.009-0.014 s of a 0.15 s run.
    void admitPairs(const ViewId* s, const ViewId* d, size_t n, uint8_t* admit) {
        std::unique_lock<std::shared_mutex> l(mu);
        for (size_t i = 0; i < n; ++i) {
            if (i + 8 < n) {  // (the two probes of a pair start in unrelated lines of a table of 10^5 keys)
                edge_index.prefetch({s[i + 8], d[i + 8]});
                edge_index.prefetch({d[i + 8], s[i + 8]});
            }
            admit[i] = (edge_index.count({s[i], d[i]}) != 0 || edge_index.count({d[i], s[i]}) != 0) ? 0 : 1;
            if (!admit[i]) continue;
            for (const ViewId id : {s[i], d[i]}) {
                if (hasVertexUnlocked(id)) continue;
                if (id < kDenseIds) {
                    if (id >= vertex_dense.size()) vertex_dense.resize(std::max<size_t>(id + 1, 2 * vertex_dense.size()), 0);
                    vertex_dense[id] = 1;
                } else {
                    vertex_sparse.emplace(id, PoseGraphVertex(id));
                }
                ++vertex_count;
            }
        }
    }
    // is ANY of the n pairs an edge, in either direction? (one lock; the scheduler's check of a speculatively formed wave)
    bool anyEdgeBetween(const ViewId* s, const ViewId* d, size_t n) const {
        std::shared_lock<std::shared_mutex> l(mu);
        for (size_t i = 0; i < n; ++i)
            if (edge_index.count({s[i], d[i]}) != 0 || edge_index.count({d[i], s[i]}) != 0) return true;
        return false;
    }
    // both endpoints of a candidate pair (pose_graph.h addVertex twice), one lock
    void addVertexPair(ViewId a, ViewId b) {
        std::unique_lock<std::shared_mutex> l(mu);
        for (const ViewId id : {a, b}) {
            if (hasVertexUnlocked(id)) continue;
            if (id < kDenseIds) {
                if (id >= vertex_dense.size()) vertex_dense.resize(std::max<size_t>(id + 1, 2 * vertex_dense.size()), 0);
                vertex_dense[id] = 1;
            } else {
                vertex_sparse.emplace(id, PoseGraphVertex(id));
            }
            ++vertex_count;
        }
    }
    // pose_graph.h:201-224: refused unless both vertices exist and the directed edge is new
    bool addEdge(ViewId s, ViewId d, const Pose& T, double score = 1.0) {
        std::unique_lock<std::shared_mutex> l(mu);
        if (!hasVertexUnlocked(s) || !hasVertexUnlocked(d)) return false;
        const uint32_t k = (uint32_t)edge_store.size();
        if (!edge_index.insert(EdgeId{s, d}, k)) return false;
        edge_store.emplace_back(s, d, T, score);
        neighboursForUpdate(s).push_back(Neighbour{d, score, k});
        neighboursForUpdate(d).push_back(Neighbour{s, score, k});
        return true;
    }
    // A wave of edges under ONE lock, each with addEdge's rule (both vertices present, id not yet there); returns how many went
    // in.  The per-vertex lists are grown once for the whole wave (at 10^5 edges and ~40 per view, one addEdge per edge spent
    // most of its 0.3 us on the lock and on re-housing those lists).
    struct NewEdge {
        ViewId src, dst;
        double score;
        const double* R;  // 9, row-major
        const double* t;  // 3
    };
    size_t addEdges(const NewEdge* items, size_t n) {
        std::unique_lock<std::shared_mutex> l(mu);
        const size_t want = edge_store.size() + n;
        if (want > edge_store.capacity()) edge_store.reserve(std::max(want, 2 * edge_store.capacity()));
        edge_index.reserve(want);
        degree_scratch.clear();
        for (size_t i = 0; i < n; ++i)
            for (ViewId v : {items[i].src, items[i].dst})
                if (v < kDenseIds) {
                    if (v >= degree_scratch.size()) degree_scratch.resize(std::max<size_t>(v + 1, 2 * degree_scratch.size()), 0);
                    ++degree_scratch[v];
                }
        if (degree_scratch.size() > adjacency_dense.size()) adjacency_dense.resize(degree_scratch.size());
        for (size_t v = 0; v < degree_scratch.size(); ++v)
            if (degree_scratch[v]) {
                std::vector<Neighbour>& nb = adjacency_dense[v];
                if (nb.size() + degree_scratch[v] > nb.capacity()) nb.reserve(std::max<size_t>(nb.size() + degree_scratch[v], 2 * nb.capacity()));
            }
        size_t added = 0;
        for (size_t i = 0; i < n; ++i) {
            const ViewId s = items[i].src, d = items[i].dst;
            if (!hasVertexUnlocked(s) || !hasVertexUnlocked(d)) continue;
            const uint32_t k = (uint32_t)edge_store.size();
            if (!edge_index.insert(EdgeId{s, d}, k)) continue;
            SE3d T;
            for (int c = 0; c < 9; ++c) T.R[c] = items[i].R[c];
            for (int c = 0; c < 3; ++c) T.t[c] = items[i].t[c];
            edge_store.emplace_back(s, d, Pose(T), items[i].score);
            neighboursForUpdate(s).push_back(Neighbour{d, items[i].score, k});
            neighboursForUpdate(d).push_back(Neighbour{s, items[i].score, k});
            ++added;
        }
        return added;
    }
    // The same with a team (round 5): `team(parts, fn)` runs fn(0) .. fn(parts - 1) concurrently and returns when all are done.
    // Which items become edges, and their positions, is decided serially (the id index: first occurrence wins, as above); the
    // 208-byte records and the adjacency entries -- the cache misses and the first touch of fresh pages -- are written by the
    // team: part p writes the records of its slice of the items and the lists of its vertices (blocks of 64 ids dealt round
    // robin: neighbouring list headers share cache lines), scanning the items in order, so every list has the order the serial insertion gives it.
    template <class Team>
    size_t addEdges(const NewEdge* items, size_t n, size_t parts, Team&& team, size_t serialBelow = 1024) {
        if (parts <= 1 || n < serialBelow) return addEdges(items, n);
        std::unique_lock<std::shared_mutex> l(mu);
        const size_t want = edge_store.size() + n;
        if (want > edge_store.capacity()) edge_store.reserve(std::max(want, 2 * edge_store.capacity()));
        edge_index.reserve(want);
        degree_scratch.clear();
        for (size_t i = 0; i < n; ++i)
            for (ViewId v : {items[i].src, items[i].dst})
                if (v < kDenseIds) {
                    if (v >= degree_scratch.size()) degree_scratch.resize(std::max<size_t>(v + 1, 2 * degree_scratch.size()), 0);
                    ++degree_scratch[v];
                }
        if (degree_scratch.size() > adjacency_dense.size()) adjacency_dense.resize(degree_scratch.size());
        slot_scratch.resize(n);
        const uint32_t base = (uint32_t)edge_store.size();
        uint32_t added = 0;
        bool sparse = false;
        for (size_t i = 0; i < n; ++i) {
            const ViewId s = items[i].src, d = items[i].dst;
            slot_scratch[i] = EdgeIndex::npos;
            if (!hasVertexUnlocked(s) || !hasVertexUnlocked(d)) continue;
            if (!edge_index.insert(EdgeId{s, d}, base + added)) continue;
            slot_scratch[i] = base + added++;
            sparse = sparse || s >= kDenseIds || d >= kDenseIds;
        }
        PoseGraphEdge* fresh = edge_store.extend(added) - base;  // fresh[k] is record k
        const uint32_t* slot = slot_scratch.data();
        team(parts, [&](size_t p) {
            for (size_t v = 0; v < degree_scratch.size(); ++v)
                if (degree_scratch[v] && (v >> 6) % parts == p) {
                    std::vector<Neighbour>& nb = adjacency_dense[v];
                    if (nb.size() + degree_scratch[v] > nb.capacity()) nb.reserve(std::max<size_t>(nb.size() + degree_scratch[v], 2 * nb.capacity()));
                }
            for (size_t i = n * p / parts; i < n * (p + 1) / parts; ++i) {
                if (slot[i] == EdgeIndex::npos) continue;
                SE3d T;
                for (int c = 0; c < 9; ++c) T.R[c] = items[i].R[c];
                for (int c = 0; c < 3; ++c) T.t[c] = items[i].t[c];
                new (fresh + slot[i]) PoseGraphEdge(items[i].src, items[i].dst, Pose(T), items[i].score);
            }
            for (size_t i = 0; i < n; ++i) {
                if (slot[i] == EdgeIndex::npos) continue;
                const ViewId s = items[i].src, d = items[i].dst;
                if (s < kDenseIds && (s >> 6) % parts == p) adjacency_dense[s].push_back(Neighbour{d, items[i].score, slot[i]});
                if (d < kDenseIds && (d >> 6) % parts == p) adjacency_dense[d].push_back(Neighbour{s, items[i].score, slot[i]});
            }
        });
        if (sparse)
            for (size_t i = 0; i < n; ++i) {
                if (slot[i] == EdgeIndex::npos) continue;
                const ViewId s = items[i].src, d = items[i].dst;
                if (s >= kDenseIds) adjacency_sparse[s].push_back(Neighbour{d, items[i].score, slot[i]});
                if (d >= kDenseIds) adjacency_sparse[d].push_back(Neighbour{s, items[i].score, slot[i]});
            }
        return added;
    }
    // room for n more edges (one rehash instead of a dozen while a wave of 10^4 edges is committed)
    void reserveEdges(size_t n) {
        std::unique_lock<std::shared_mutex> l(mu);
        const size_t want = edge_store.size() + n;  // geometric: a reserve per wave must not re-house the graph every time
        if (want > edge_store.capacity()) edge_store.reserve(std::max(want, 2 * edge_store.capacity()));
        edge_index.reserve(want > edge_store.capacity() ? want : edge_store.capacity());
    }
    PoseGraphEdge getEdgeById(const EdgeId& id) const {
        std::shared_lock<std::shared_mutex> l(mu);
        const uint32_t k = edge_index.find(id);
        return k == EdgeIndex::npos ? PoseGraphEdge() : edge_store[k];
    }
    // the edges first .. last - 1 in insertion order, by reference and under ONE shared lock: fn(position, edge)
    template <class Fn>
    void forEachEdge(size_t first, size_t last, Fn fn) const {
        std::shared_lock<std::shared_mutex> l(mu);
        for (size_t k = first; k < std::min(last, edge_store.size()); ++k) fn(k, edge_store[k]);
    }
    std::vector<EdgeId> getEdgeIds() const {  // insertion order, like the reference's edges_ids
        std::shared_lock<std::shared_mutex> l(mu);
        std::vector<EdgeId> ids(edge_store.size());
        for (size_t k = 0; k < ids.size(); ++k) ids[k] = {edge_store[k].getSourceId(), edge_store[k].getDestinationId()};
        return ids;
    }
    bool getEdgesByVertex(const ViewId& id, std::vector<EdgeId>& out) const {  // pose_graph.h:139-151
        std::shared_lock<std::shared_mutex> l(mu);
        const std::vector<Neighbour>* nb = neighboursOf(id);
        if (!nb) return false;
        out.resize(nb->size());
        for (size_t k = 0; k < out.size(); ++k) {
            const PoseGraphEdge& e = edge_store[(*nb)[k].edge];
            out[k] = {e.getSourceId(), e.getDestinationId()};
        }
        return true;
    }
    // the edges of a vertex in insertion order, by reference and under ONE shared lock (the traversal's inner loop:
    // getEdgesByVertex + getEdgeById per edge copy an id list and a Pose per visit)
    template <class Fn>
    bool forEachEdgeOf(const ViewId& id, Fn fn) const {
        std::shared_lock<std::shared_mutex> l(mu);
        return forEachEdgeOfFrozen(id, fn);
    }
    // The same WITHOUT the lock, for callers that know no writer runs meanwhile -- the wave scheduler searches the paths of
    // a whole wave on kCoreNumber threads between two commits (the shared lock's counter is one cache line that twenty
    // readers would pass around once per visited vertex).
    template <class Fn>
    bool forEachEdgeOfFrozen(const ViewId& id, Fn fn) const {
        const std::vector<Neighbour>* nb = neighboursOf(id);
        if (!nb) return false;
        for (const Neighbour& n : *nb) fn(edge_store[n.edge]);
        return true;
    }
    // What the A* expansion reads of an edge -- the vertex at its other end and its score -- from a compact per-vertex
    // array (24 bytes per entry, contiguous) instead of the 200-byte edge records scattered over the store.  No lock.
    template <class Fn>
    bool forEachNeighbourFrozen(const ViewId& id, Fn fn) const {
        const std::vector<Neighbour>* nb = neighboursOf(id);
        if (!nb) return false;
        for (const Neighbour& n : *nb) fn(n.other, n.score);
        return true;
    }
    const PoseGraphEdge* findEdgeFrozen(const EdgeId& id) const {  // (no lock: see forEachEdgeOfFrozen)
        const uint32_t k = edge_index.find(id);
        return k == EdgeIndex::npos ? nullptr : &edge_store[k];
    }
    size_t getEdgeNumberByVertex(const ViewId& id) const {
        std::shared_lock<std::shared_mutex> l(mu);
        const std::vector<Neighbour>* nb = neighboursOf(id);
        return nb ? nb->size() : 0;
    }

   protected:
    struct EdgeIdHash {
        size_t operator()(const EdgeId& e) const noexcept {
            uint64_t h = (uint64_t)e.first * 0x9E3779B97F4A7C15ull ^ ((uint64_t)e.second + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
            return (size_t)(h ^ (h >> 29));
        }
    };
    struct Neighbour { ViewId other; double score; uint32_t edge; };  // edge = position in edge_store
    // View ids are image indices (types.h:9-14): ids below 2^22 index plain arrays, anything larger falls back to hash maps.
    static constexpr ViewId kDenseIds = (ViewId)1 << 22;
    bool hasVertexUnlocked(ViewId id) const {
        if (id < kDenseIds) return id < vertex_dense.size() && vertex_dense[id] != 0;
        return vertex_sparse.count(id) != 0;
    }
    const std::vector<Neighbour>* neighboursOf(ViewId id) const {  // nullptr while the vertex has no edge
        if (id < kDenseIds) return id < adjacency_dense.size() && !adjacency_dense[id].empty() ? &adjacency_dense[id] : nullptr;
        auto it = adjacency_sparse.find(id);
        return it == adjacency_sparse.end() ? nullptr : &it->second;
    }
    std::vector<Neighbour>& neighboursForUpdate(ViewId id) {
        if (id >= kDenseIds) return adjacency_sparse[id];
        if (id >= adjacency_dense.size()) adjacency_dense.resize(std::max<size_t>(id + 1, 2 * adjacency_dense.size()));
        return adjacency_dense[id];
    }
    mutable std::shared_mutex mu;
    size_t vertex_count = 0;
    std::vector<uint8_t> vertex_dense;
    std::unordered_map<ViewId, PoseGraphVertex> vertex_sparse;
    EdgeArray edge_store;                                          // insertion order
    // (src, dst) -> position in edge_store.  Ids below 2^32 (every real case) live in an open-addressing table of packed
    // 64-bit keys -- no node allocation per edge, one probe sequence in one array; anything larger in a std hash map.
    class EdgeIndex {
       public:
        static constexpr uint32_t npos = 0xFFFFFFFFu;
        uint32_t find(const EdgeId& id) const {
            if (inBig(id)) {
                auto it = big.find(id);
                return it == big.end() ? npos : it->second;
            }
            if (keys.empty()) return npos;
            const uint64_t k = pack(id);
            for (size_t i = slot(k);; i = (i + 1) & (keys.size() - 1)) {
                if (keys[i] == k) return vals[i];
                if (keys[i] == kEmpty) return npos;
            }
        }
        size_t count(const EdgeId& id) const { return find(id) != npos; }
        void prefetch(const EdgeId& id) const {
            if (!keys.empty() && !inBig(id)) __builtin_prefetch(&keys[slot(pack(id))]);
        }
        bool insert(const EdgeId& id, uint32_t v) {  // false if the id is already there
            if (inBig(id)) return big.emplace(id, v).second;
            if ((used + 1) * 2 > keys.size()) grow(std::max<size_t>(64, 2 * keys.size()));
            const uint64_t k = pack(id);
            for (size_t i = slot(k);; i = (i + 1) & (keys.size() - 1)) {
                if (keys[i] == k) return false;
                if (keys[i] == kEmpty) {
                    keys[i] = k;
                    vals[i] = v;
                    ++used;
                    return true;
                }
            }
        }
        void reserve(size_t n) {
            size_t cap = 64;
            while (cap < 2 * n) cap *= 2;
            if (cap > keys.size()) grow(cap);
        }

       private:
        static constexpr uint64_t kEmpty = ~0ull;  // the one pair that packs to it, (2^32 - 1, 2^32 - 1), lives in `big`
        static bool inBig(const EdgeId& id) { return ((id.first | id.second) >> 32) != 0 || pack(id) == kEmpty; }
        static uint64_t pack(const EdgeId& id) { return ((uint64_t)id.first << 32) | (uint64_t)id.second; }
        size_t slot(uint64_t k) const {
            k *= 0x9E3779B97F4A7C15ull;
            return (size_t)(k ^ (k >> 29)) & (keys.size() - 1);
        }
        void grow(size_t cap) {
            std::vector<uint64_t> ok(cap, kEmpty);
            std::vector<uint32_t> ov(cap, 0);
            ok.swap(keys);
            ov.swap(vals);
            used = 0;
            for (size_t i = 0; i < ok.size(); ++i)
                if (ok[i] != kEmpty) {
                    for (size_t j = slot(ok[i]);; j = (j + 1) & (keys.size() - 1))
                        if (keys[j] == kEmpty) {
                            keys[j] = ok[i];
                            vals[j] = ov[i];
                            ++used;
                            break;
                        }
                }
        }
        std::vector<uint64_t> keys;
        std::vector<uint32_t> vals;
        size_t used = 0;
        std::unordered_map<EdgeId, uint32_t, EdgeIdHash> big;
    };
    EdgeIndex edge_index;
    std::vector<std::vector<Neighbour>> adjacency_dense;           // per vertex, insertion order
    std::vector<uint32_t> degree_scratch;                          // addEdges: new edges per vertex of the wave
    std::vector<uint32_t> slot_scratch;                            // addEdges with a team: the position each item's edge takes
    std::unordered_map<ViewId, std::vector<Neighbour>> adjacency_sparse;
};

// ---- the engine handle shared by the classes below ---------------------------------------------------
class PgiError : public std::runtime_error {
   public:
    using std::runtime_error::runtime_error;
};

class Engine {
   public:
    explicit Engine(int device = -1, const pgi_params* prm = nullptr) : ctx(pgi_create(device, prm)) {
        if (!ctx) throw PgiError(std::string("pgi_create: ") + pgi_last_error());
    }
    ~Engine() { pgi_destroy(ctx); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    pgi_ctx* get() const { return ctx; }
    static void check(int rc) {
        if (rc < 0) throw PgiError(pgi_last_error());
    }

   protected:
    pgi_ctx* ctx;
};


class EssentialMatrixEvaluator {  // include/graph_traversal.h:82-170
   public:
    explicit EssentialMatrixEvaluator(Engine& e) : eng(&e) {}
    // graph_traversal.h:136-168.  NOTE the reference compares the SQUARED residual with the un-squared
    // kThreshold_ (line 164); this call reproduces that: pass 1.5*thr exactly as estimatePose does (:985-989).
    void getInliers(const CorrespondenceMatrix& kCorrespondences_, const Matrix3d& kDescriptor_,
                    const double& kThreshold_, std::vector<size_t>& inliers_) const {
        std::vector<uchar> mask((size_t)kCorrespondences_.rows);
        uint32_t cnt = 0;
        uchar none = 0;
        // one re-entrant call on host pointers (private slot of the context: no allocation, no shared stream)
        Engine::check(pgi_score_pose_f64_host(eng->get(), kCorrespondences_.ptr(), (uint32_t)kCorrespondences_.rows,
                                              kDescriptor_.data(), kThreshold_, 0, &cnt, mask.empty() ? &none : mask.data()));
        inliers_.reserve((size_t)kCorrespondences_.rows);
        for (size_t i = 0; i < mask.size(); ++i)
            if (mask[i]) inliers_.emplace_back(i);
    }

   protected:
    Engine* eng;
};

template <typename _Evaluator = EssentialMatrixEvaluator>
class InTraversalPoseTester {  // include/graph_traversal.h:174-242
   public:
    InTraversalPoseTester(Engine& e, double kInlierOutlierThreshold_, size_t kMinimumInlierNumber_,
                          const CorrespondenceMatrix* correspondences_)
        : eng(&e),
          kSquaredInlierOutlierThreshold(kInlierOutlierThreshold_ * kInlierOutlierThreshold_),
          kMinimumInlierNumber(kMinimumInlierNumber_),
          correspondences(correspondences_) {}
    // graph_traversal.h:194-233: true as soon as kMinimumInlierNumber inliers exist; inlierNumber_ is then
    // exactly kMinimumInlierNumber (the reference returns at that inlier), otherwise the full count.
    bool test(const SE3d& kPose_, size_t& inlierNumber_) const {
        if (!kMinimumInlierNumber) {  // nothing to reach
            inlierNumber_ = 0;
            return true;
        }
        const Matrix3d E = pose::getEssentialMatrixFromRelativePose(kPose_);
        uint32_t cnt = 0;
        const int rc = pgi_score_pose_f64_host(eng->get(), correspondences->ptr(), (uint32_t)correspondences->rows, E.data(),
                                               kSquaredInlierOutlierThreshold, (uint32_t)kMinimumInlierNumber, &cnt, nullptr);
        Engine::check(rc);  // the device scan stops at that inlier too (graph_traversal.h:221-225)
        inlierNumber_ = cnt;
        return rc == 1;
    }

   protected:
    Engine* eng;
    double kSquaredInlierOutlierThreshold;
    size_t kMinimumInlierNumber;
    const CorrespondenceMatrix* correspondences;
};

class PoseGraphBuilder {  // include/pose_graph_builder.h:25-171
   public:
    PoseGraphBuilder(const size_t kCoreNumber_, const size_t kMaximumTrackletNumber_,
                     const size_t kMaximumSearchDepth_, const size_t kMaximumPathNumber_,
                     const size_t kMinimumInlierNumber_, const size_t kMinimumPointNumber_,
                     const size_t kMaximumPointNumberForEpipolarHashing_, const double kTraversalHeuristicsWeight_,
                     const double kSimilarityThreshold_, const double kInlierOutlierThreshold_,
                     const std::string& kImagePath_, const std::string& kWorkspacePath_,
                     const std::string& kSimilarityGraphPath_, const std::string& kFocalLengthPath_,
                     const bool kUsePathFinding_, const bool kUseGPU_, const bool kUseEpipolarHashing_)
        : kCoreNumber(kCoreNumber_),
          kMinimumInlierNumber(kMinimumInlierNumber_),
          kMinimumPointNumber(kMinimumPointNumber_),
          kMaximumPointNumberForEpipolarHashing(kMaximumPointNumberForEpipolarHashing_),
          kMaximumSearchDepth(kMaximumSearchDepth_),
          kMaximumPathNumber(kMaximumPathNumber_),
          kMaximumTrackletNumber(kMaximumTrackletNumber_),
          kImagePath(kImagePath_),
          kWorkspacePath(kWorkspacePath_),
          kSimilarityGraphPath(kSimilarityGraphPath_),
          kFocalLengthPath(kFocalLengthPath_),
          kUseGPU(kUseGPU_),
          kUsePathFinding(kUsePathFinding_),
          kUseEpipolarHashing(kUseEpipolarHashing_),
          kTraversalHeuristicsWeight(kTraversalHeuristicsWeight_),
          kInlierOutlierThreshold(kInlierOutlierThreshold_),
          kSimilarityThreshold(kSimilarityThreshold_) {
        pgi_params p;
        pgi_default_params(&p);
        p.min_inliers = (uint32_t)kMinimumInlierNumber_;
        engine.reset(new Engine(-1, &p));
    }

    // A candidate view pair with its normalised correspondences (what processImages hands to
    // estimatePose, pose_graph_builder.h:553-565, 616-627).
    struct ViewPair {
        ViewId src, dst;
        double similarity;
        CorrespondenceMatrix correspondences;
        double normalizedThreshold;
        std::vector<SE3d> poseGuesses;  // from the A* traversal (0 or 1 in the reference)
    };

    // The estimatePose seam (pose_graph_builder.h:153-164, 940-1078).  reconstruction_, the pixel threshold
    // and the view indices are unused by the reference's body and are therefore not parameters here.
    bool estimatePose(const size_t kMinimumInlierNumber_, const CorrespondenceMatrix& kCorrespondences_,
                      const double kThreshold_, const std::vector<SE3d>& poseGuesses_, SE3d& estimatedPose_,
                      std::vector<uchar>& inlierMask_, size_t& inlierNumber_, uint64_t seed = 0,
                      uint64_t pairId = 0) {
        std::vector<double> g(12 * poseGuesses_.size());
        for (size_t i = 0; i < poseGuesses_.size(); ++i) {
            for (int c = 0; c < 9; ++c) g[12 * i + c] = poseGuesses_[i].R[c];
            for (int c = 0; c < 3; ++c) g[12 * i + 9 + c] = poseGuesses_[i].t[c];
        }
        inlierMask_.assign((size_t)kCorrespondences_.rows, 0);  // :1000, :1034
        pgi_edge e;
        uchar dummy = 0;
        const int rc = pgi_estimate_pose(engine->get(), kCorrespondences_.ptr(), (uint32_t)kCorrespondences_.rows,
                                         kThreshold_, g.empty() ? nullptr : g.data(), (uint32_t)poseGuesses_.size(),
                                         (uint32_t)kMinimumInlierNumber_, seed, pairId, &e,
                                         inlierMask_.empty() ? &dummy : inlierMask_.data());
        Engine::check(rc);
        inlierNumber_ = e.n_inl;
        if (rc != 1) return false;  // :1053-1054, :1069-1070
        for (int c = 0; c < 9; ++c) estimatedPose_.R[c] = e.R[c];
        for (int c = 0; c < 3; ++c) estimatedPose_.t[c] = e.t[c];
        return true;
    }

    // Batched form used by run(): every pair of a wave in one launch; edges with score =
    // inliers / matches are added to the pose graph (pose_graph_builder.h:645-654).  A pair's LAST
    // poseGuess (if any) is first screened like InTraversalPoseTester::test does inside the A* traversal
    // (>= 5 rows with squared Sampson distance < (1.5 thr)^2; pose_graph_builder.h:798-811) -- all
    // screens of the wave in one score launch -- and only accepted guesses reach estimatePose.
    //
    // Multi-GPU (setHostComm + dist::attach): the pairs are cut into world contiguous, row-balanced blocks
    // (dist::shardBounds); this rank uploads and estimates only its block (seeds are keyed by the pair's index in
    // `pairs`, so a block reproduces the single-process result bit for bit), the 200-byte edge records are
    // all-gathered (pgi_allgather_edges: RCCL over xGMI, or the host transport when ranks share a device) and every
    // rank commits the identical full table.  d_edges_out (optional): receives the gathered table, still in HBM
    // (pairs.size() records; caller-allocated), e.g. to feed pgi_rotation_average_edges without a host round trip.
    // prepareGuesses (optional): called on the host with ranges [first, last) of `pairs` -- this rank's launch groups, in
    // order -- before the range's rows are staged; it may fill poseGuesses of exactly those pairs (the scheduler runs its A*
    // searches there, so that they overlap the device's work on the previous group).
    // deferredInsertion (optional, needs edges_out): the edges are NOT put into poseGraph_; the function receives the closure
    // that does it (returns the number of edges; reads *edges_out, pairs and poseGraph_, which must outlive it) -- for a caller
    // with device work of its own to run meanwhile (estimateAndAverage).  The return value is 0 then.
    size_t estimatePoses(const std::vector<ViewPair>& pairs, PoseGraph& poseGraph_, uint64_t seed = 0,
                         std::vector<pgi_edge>* edges_out = nullptr, bool screenGuesses = false,
                         pgi_edge* d_edges_out = nullptr,
                         const std::function<void(size_t, size_t)>* prepareGuesses = nullptr,
                         std::function<size_t()>* deferredInsertion = nullptr,
                         const std::function<const std::vector<ViewPair>*()>* nextWave = nullptr, bool rowsOnly = false);
    // nextWave (optional; the scheduler's): called once while this call's kernels run; returns the pairs of the NEXT call (or
    // nullptr) -- their rows are converted and uploaded into the second staging block meanwhile, and the call that later brings
    // exactly that vector finds them resident.  rowsOnly: that prefetch itself (internal).

    // BASELINE config 4: shard -> estimate -> gather -> replicated L1/IRLS rotation averaging; the gathered records
    // stay in HBM between the exchange and the solve.  R_i are world->camera, one per view id < numViews.
    struct GlobalRotations {
        std::vector<Matrix3d> rotations;
        uint32_t iterations = 0, edgesUsed = 0;
    };
    GlobalRotations estimateAndAverage(const std::vector<ViewPair>& pairs, PoseGraph& poseGraph_, size_t numViews,
                                       uint64_t seed = 0, std::vector<pgi_edge>* edges_out = nullptr,
                                       const pgi_rotavg_params* rotavgParams = nullptr);
    // Rotation averaging over the edges already in a pose graph (weights = edge scores), e.g. after run().
    GlobalRotations averageRotations(const PoseGraph& poseGraph_, size_t numViews,
                                     const pgi_rotavg_params* rotavgParams = nullptr);

    struct RunStatistics {  // the reference's RunningStatistics keys that concern this path (utils.h:15-73)
        size_t pairsProcessed = 0, edgesAdded = 0, pathsSearched = 0, pathsFound = 0, touchedNodes = 0,
               posesFromGuess = 0, hypotheses = 0, waves = 0;
        // accepted A* guesses whose inlier count under the SQUARED bound (1.5 thr)^2 is below kMinimumInlierNumber: edges
        // that exist only through the reference's un-squared getInliers bound (graph_traversal.h:149,164)
        size_t quirkOnlyGuesses = 0;
    };

    // Wave-scheduled run over caller-provided candidate pairs (the image / feature / matching stages of
    // the reference's run() -- pose_graph_builder.h:173-239, 352-715 -- are outside this build's scope,
    // DESIGN.md §6): pairs above kSimilarityThreshold with at least kMinimumPointNumber matches, in
    // descending similarity (imagesimilarity_graph.h max-heap order), are estimated in batches.  With
    // kUsePathFinding and a similarity table, pairs already connected in the graph (VisibilityTable) get
    // an A* pose guess computed on the graph committed by the previous waves (findPath, :785-862).
    // Multi-GPU (BASELINE config 5, SURVEY §8e wave protocol): every rank walks the same sorted candidate list and
    // forms the same wave; a rank runs host A* on the committed snapshot only for the pairs of its share, estimates
    // them, the records are all-gathered and every rank commits the whole wave in wave order -- graph, visibility
    // and statistics are identical on all ranks and identical to the single-process run.
    // seedBase: wave w draws its hypotheses with seed seedBase + w (pgi_batch.seed).
    RunStatistics run(std::vector<ViewPair>& candidatePairs, PoseGraph& poseGraph_, size_t waveSize = 4096,
                      const class SimilarityTable* similarityTable = nullptr, uint64_t seedBase = 0);

    // The outer API of the reference: void run(Reconstruction&, PoseGraph&) (pose_graph_builder.h:69-71, 173-239) on
    // the workspace named by the constructor's paths (list_with_focals.txt, similarity matrix, image_data.h5,
    // keypoints.h5, correspondences.h5).  Defined in host/hdf5_cache.hpp (needs the HDF5 C library).
    void run(Reconstruction& reconstruction_, PoseGraph& poseGraph_);
    // pose_graph_builder.h:241-291
    void initializeReconstruction(const size_t& kImageNumber_,
                                  const std::vector<std::tuple<std::string, double, double, double>>& imageData_,
                                  Reconstruction& reconstruction_, PoseGraph& poseGraph_);

    // ---- the loop body of processImages on in-memory features (pose_graph_builder.h:391-709 minus file I/O) ----
    // What loadFeatures returns for one image (:463-482) plus its pinhole camera K = [f 0 w/2; 0 f h/2; 0 0 1] (:286).
    struct ViewFeatures {
        std::vector<float> keypoints;    // n x 2 pixel coordinates (cv::KeyPoint::pt)
        std::vector<float> descriptors;  // n x 128 RootSIFT, row-major
        double focalLength = 1.0, width = 0.0, height = 0.0;
        size_t size() const { return keypoints.size() / 2; }
    };
    struct ViewFeaturesRef {  // the same, not owning (arrays of a caller in another language: pgih_run_features)
        const float* keypoints = nullptr;
        const float* descriptors = nullptr;
        size_t n = 0;
        double focalLength = 1.0, width = 0.0, height = 0.0;
        size_t size() const { return n; }
    };
    struct CandidatePair {
        ViewId src, dst;
        double similarity;
    };
    struct FeatureRunStatistics : RunStatistics {  // the "[Matching]", "[Quick matching]", "[Epipolar Hashing]" keys
        size_t matchingRuns = 0, quickMatchingRuns = 0, guidedMatchingRuns = 0, guidedMatchesAdded = 0, trackNumber = 0,
               tooFewMatches = 0, cachedMatchLoads = 0;
        // wall-clock seconds per stage (RunningStatistics' "[Matching]", "[Quick matching]", "[A*]", "[Pose estimation]",
        // "[Epipolar Hashing]" timers of the reference)
        double secQuickMatching = 0, secMatching = 0, secCorrespondences = 0, secAStar = 0, secPoseEstimation = 0,
               secGuidedMatching = 0, secTrackUpdate = 0, secUpload = 0;  // secUpload: features -> HBM + descriptor preparation
    };
    // Per wave of candidate pairs (descending similarity): tracklet correspondences for pairs the graph already
    // connects (:493-518), descriptor matching for the rest (:521-546), createCorrespondenceMatrix (:553-565), A*
    // pose guesses (:568-599), estimatePose (:616-627), edge + visibility update (:645-654, :692), guided matching
    // and tracklet update (:657-709).  Matching, correspondence building, guess screening, pose estimation and guided
    // matching and the tracklet store are device launch sequences per wave; A* and the graph stay on the host.
    // cachedMatches (optional): the correspondences.h5 lookup of matchFeatures (feature_utils.h:113-133) -- returns
    // true and fills the (src index, dst index, ratio) list when the pair's matches are already known.
    typedef std::function<bool(ViewId, ViewId, std::vector<std::tuple<size_t, size_t, double>>&)> MatchLookup;
    FeatureRunStatistics processFeatures(const std::vector<ViewFeatures>& views, std::vector<CandidatePair>& candidatePairs,
                                         PoseGraph& poseGraph_, size_t waveSize = 1024,
                                         const class SimilarityTable* similarityTable = nullptr,
                                         const MatchLookup* cachedMatches = nullptr);
    FeatureRunStatistics processFeatures(const std::vector<ViewFeaturesRef>& views, std::vector<CandidatePair>& candidatePairs,
                                         PoseGraph& poseGraph_, size_t waveSize = 1024,
                                         const class SimilarityTable* similarityTable = nullptr,
                                         const MatchLookup* cachedMatches = nullptr);

    Engine& getEngine() { return *engine; }
    // the reference's `RunningStatistics statistics` (pose_graph_builder.h:385): filled by run / processFeatures with
    // the reference's key names, printed by run(Reconstruction&, PoseGraph&) like :712
    RunningStatistics& getStatistics() { return statistics; }
    // One process per GPU: keep this process on the CPUs (and, by first touch, the memory) of the NUMA node the device hangs
    // on -- call it before the inputs are read.  device < 0: the current HIP device.  Returns the node or -1 (see the .cpp).
    static int bindProcessToDeviceNode(int device = -1);
    // BASELINE config 5: use the A* chain's ROTATION only and re-estimate the translation direction on the GPU
    // (pgi_params.guess_mode = 1) instead of the reference's score -> refit of the chained pose (mode 0).  The chained
    // translation is a sum of unit baselines and therefore meaningless; with this switch on, run() hands every chained
    // pose to the estimator without the InTraversalPoseTester screen (which tests exactly that translation).
    void setRotationGuidedGuesses(bool on) {
        pgi_params p;
        Engine::check(pgi_get_params(engine->get(), &p));
        p.guess_mode = on ? 1u : 0u;
        Engine::check(pgi_set_params(engine->get(), &p));
        rotationGuidedGuesses = on;
    }
    // tracklets in HBM (pgi_tracklets_*, the default) or in the host store (host/tracklets.hpp); same results
    void setDeviceTracklets(bool on) { deviceTracklets = on; }
    // processFeatures: progressive sampling over the matcher's ratio-sorted rows (pgi_params.sampler = 1; on by default there)
    void setProgressiveSampling(bool on) { progressiveSampling = on; }
    // graph-cut local optimisation (pgi_params.lo_graph_cut = lambda * 64; 0 = off, the default): the "GC" of GC-RANSAC
    void setGraphCutLocalOptimisation(uint32_t lambda64) {
        pgi_params p;
        Engine::check(pgi_get_params(engine->get(), &p));
        p.lo_graph_cut = lambda64;
        Engine::check(pgi_set_params(engine->get(), &p));
    }
    // one process per GPU: the host-side channel of this rank (also install the engine's transport: dist::attach)
    void setHostComm(dist::HostComm* comm) { hostComm = comm; }
    uint32_t worldSize() const;
    uint32_t worldRank() const;

   protected:
    void warnQuirkOnlyGuesses(size_t quirkOnly, size_t acceptedGuesses);
    RunningStatistics statistics;
    dist::HostComm* hostComm = nullptr;
    bool rotationGuidedGuesses = false;
    bool deviceTracklets = true;
    bool progressiveSampling = true;
    const size_t kCoreNumber, kMinimumInlierNumber, kMinimumPointNumber, kMaximumPointNumberForEpipolarHashing,
        kMaximumSearchDepth, kMaximumPathNumber, kMaximumTrackletNumber;
    const std::string kImagePath, kWorkspacePath, kSimilarityGraphPath, kFocalLengthPath;
    const bool kUseGPU, kUsePathFinding, kUseEpipolarHashing;
    const double kTraversalHeuristicsWeight, kInlierOutlierThreshold, kSimilarityThreshold;
    std::unique_ptr<Engine> engine;
    // grow-only staging of estimatePoses: one page-locked host block and one device block, reused from wave to wave
    // (defined in the implementation file: HIP stays out of this header)
    struct Staging;
    std::shared_ptr<Staging> staging;
    double* rowsOnlyConvertSeconds = nullptr;  // where a prefetch (helper thread) leaves its conversion time for the calling thread
    uint64_t lastQuirkOnlyGuesses = 0;  // of the last estimatePoses call (all ranks)
};

namespace pose {
// pose_utils.h:172-252: returns the vote count of the chosen candidate
int getPoseFromEssentialMatrix(Engine& eng, const Matrix3d& essential_matrix_,
                               const CorrespondenceMatrix& normalized_correspondences_, Matrix3d& rotation_,
                               Vector3d& translation_);
}  // namespace pose

}  // namespace reconstruction
