// graph_traversal.hpp -- host-side A* pose-guess search over the pose graph (no GPU dependency).
//
// Mirrors, with the reference's names:
//   SimilarityTable                imagesimilarity_graph.h:15-105 (dense N x N similarity, max-heap of pairs >= threshold)
//   VisibilityTable                visibility_table.h:12-171 -> union-find (the reference's closure is buggy)
//   ImageSimilarityHeuristics      graph_traversal.h:569-596
//   CostComparator / AStarTraversal::getPath   graph_traversal.h:598-870
//   PoseGraphTraversal::recoverPath            graph_traversal.h:290-348
// as configured by PoseGraphBuilder::findPath (pose_graph_builder.h:785-862): multi-path mode (no 'Seen' marking),
// minimum inlier ratio 0, depth cap kMaximumSearchDepth, and at most ONE recovered path (:799, graph_traversal.h:799-800).
// Heap ties on the combined weight are broken by insertion order (the reference leaves them to the STL).
// The in-traversal pose test (graph_traversal.h:787-797) is a callback so that the GPU scheduler can batch
// the tests of a whole wave after the searches (equivalent: the search ends at the first recovered path).
#pragma once
#include <algorithm>
#include <fstream>
#include <cstdlib>
#include <iterator>
#include <functional>
#include <sstream>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

#include "pose_graph_builder.hpp"

namespace reconstruction {

class SimilarityTable {
   public:
    SimilarityTable(size_t num_imgs, double image_similarity_threshold_, bool build_priority_queue_ = true)
        : similarity(num_imgs, std::vector<double>(num_imgs, 1.0)),  // the reference initialises to 1.0f (:55-65)
          size(num_imgs),
          build_priority_queue(build_priority_queue_),
          image_similarity_threshold(image_similarity_threshold_) {}
    bool setSimilarity(ViewId from, ViewId to, double val) {
        if (from >= size || to >= size) return false;
        if (build_priority_queue && from != to && image_similarity_threshold <= val) {
            views.insert(from);
            views.insert(to);
            view_pair_queue.emplace(val, from, to);
        }
        similarity[from][to] = val;
        similarity[to][from] = val;
        return true;
    }
    double getSimilarity(ViewId from, ViewId to) const {
        if (from >= size || to >= size) return -1.0;  // NO_SUCH_VERTEX
        return similarity[from][to];
    }
    // The similarity matrix file (format: imagesimilarity_graph.h:108-171): N lines of N numbers, "%1.3f" as written by
    // get_image_similarity.py.  The whole file is tokenised in one pass with strtod and checked strictly -- exactly N x N
    // numbers, N per line, nothing else (trailing blank lines are tolerated) -- before a single cell is touched, so a
    // bad file leaves the table as it was.
    // Heap-entry rule, kept because the candidate order depends on it: cells are written in row-major order, and pair
    // (i, j) is queued when i != j, the value reaches the threshold and it DIFFERS from what the mirrored cell (j, i)
    // holds at that moment -- the table's previous content (1.0 after construction) while j > i, the file's own value
    // once row j has been written.  For a symmetric file that is: each pair once, as (i < j), and never a similarity of
    // exactly 1.0.
    bool loadFromFile(const std::string& fname) {
        std::ifstream file(fname, std::ios::binary);
        if (!file.is_open()) return false;
        const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        const size_t N = similarity.size();
        std::vector<double> cell;
        cell.reserve(N * N);
        const char* p = text.c_str();
        const char* const end = p + text.size();
        size_t in_line = 0;
        while (p < end) {
            const char c = *p;
            if (c == '\n') {
                if (in_line != 0 && in_line != N) return false;   // a short or long row
                if (in_line == 0 && cell.size() != N * N && !cell.empty()) return false;  // blank line inside the matrix
                in_line = 0;
                ++p;
            } else if (c == ' ' || c == '\t' || c == '\r') {
                ++p;
            } else {
                char* q = nullptr;
                const double v = std::strtod(p, &q);
                if (q == p || cell.size() == N * N) return false;  // not a number / more than N x N values
                cell.push_back(v);
                ++in_line;
                p = q;
            }
        }
        if ((in_line != 0 && in_line != N) || cell.size() != N * N) return false;
        for (size_t k = 0; k < N * N; ++k) {
            const size_t i = k / N, j = k % N;
            const double v = cell[k];
            const double mirrored = similarity[j][i];  // previous content for j > i, this file's value for j < i
            similarity[i][j] = v;
            if (build_priority_queue && i != j && image_similarity_threshold <= v && mirrored != v) {
                views.insert(i);
                views.insert(j);
                view_pair_queue.emplace(v, i, j);
            }
        }
        return true;
    }
    std::priority_queue<std::tuple<double, ViewId, ViewId>>& getMutablePrioritizedViewPairs() { return view_pair_queue; }
    const std::unordered_set<ViewId>& getKeptViews() const { return views; }

   protected:
    std::vector<std::vector<double>> similarity;
    size_t size;
    const bool build_priority_queue;
    const double image_similarity_threshold;
    std::priority_queue<std::tuple<double, ViewId, ViewId>> view_pair_queue;
    std::unordered_set<ViewId> views;
};

class VisibilityTable {  // "is there already a path between two views?" (pose_graph_builder.h:456-457, 692)
   public:
    explicit VisibilityTable(size_t n) : parent(n) {
        for (size_t i = 0; i < n; ++i) parent[i] = i;
    }
    bool hasLink(ViewId a, ViewId b) const { return find(a) == find(b); }
    void addLink(ViewId a, ViewId b) {
        a = find(a);
        b = find(b);
        if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }

   protected:
    ViewId find(ViewId a) const {
        while (parent[a] != a) a = parent[a];
        return a;
    }
    std::vector<ViewId> parent;
};

class ImageSimilarityHeuristics {  // graph_traversal.h:569-596
   public:
    explicit ImageSimilarityHeuristics(const SimilarityTable& t) : kSimilarityTable(t) {}
    static const char* name() { return "image similarity-based"; }
    double getCost(const ViewId& a, const ViewId& b) const {
        return std::clamp(kSimilarityTable.getSimilarity(a, b), 0.0, 1.0);
    }

   protected:
    const SimilarityTable& kSimilarityTable;
};

// graph_traversal.h:290-348: pose <- T_edge * pose, or T_edge^-1 * pose for a reversed edge
inline bool recoverPath(const PoseGraph& g, const std::vector<ViewId>& path, SE3d& pose_) {
    pose_ = SE3d();
    for (size_t i = 1; i < path.size(); ++i) {
        const ViewId a = path[i - 1], b = path[i];
        if (g.hasEdge(a, b))
            pose_ = g.getEdgeById({a, b}).getValue().getTransform() * pose_;
        else if (g.hasEdge(b, a))
            pose_ = g.getEdgeById({b, a}).getValue().getTransform().inverse() * pose_;
        else
            return false;  // "This should never happen." (:332-338)
    }
    return true;
}

template <typename _NodeCostHeuristics>
class AStarTraversal {
   public:
    using PoseTest = std::function<bool(const SE3d&)>;  // InTraversalPoseTester::test, or empty
    AStarTraversal(const PoseGraph* kPoseGraph_, const _NodeCostHeuristics& kHeuristicsObject_, double weight_,
                   double kMinimumInlierRatio_ = 0.0, size_t kMaximumDepth_ = (size_t)-1)
        : kPoseGraph(kPoseGraph_),
          kHeuristicsObject(kHeuristicsObject_),
          weight(weight_),
          kMinimumInlierRatio(kMinimumInlierRatio_),
          kMaximumDepth(kMaximumDepth_) {}
    static constexpr const char* name() { return "a-star"; }

    // graph_traversal.h:679-870.  path_ / poses_ receive the (single) recovered path and its chained pose;
    // with a pose test the pose is returned only if the test accepts it (:787-797).
    void getPath(const ViewId kFrom_, const ViewId kTo_, std::vector<ViewId>& path_, std::vector<SE3d>& poses_,
                 size_t& touchedNodes_, size_t& foundPaths_, bool& pathExists_, const PoseTest& test = PoseTest()) const {
        // (the reference copies the parent list into every open node; here an open node points at a link of a shared chain,
        //  and the expanded set is a flat list -- the same search, without an allocation per pushed node)
        struct Link { ViewId vertex; int prev; };
        struct Node {
            double edgeCost, nextCost, combined;
            size_t seq;
            ViewId vertex;
            int chain;  // link of the parent vertex, -1 at the start vertex
            size_t depth;
        };
        auto worse = [](const Node& a, const Node& b) {  // max-heap on combined, earlier insertion first
            if (a.combined != b.combined) return a.combined < b.combined;
            return a.seq > b.seq;
        };
        std::priority_queue<Node, std::vector<Node>, decltype(worse)> openNodes(worse);
        std::vector<Link> chain;
        std::vector<ViewId> nodeStates;  // vertices that have been expanded (Open/Closed)
        size_t seq = 0;
        openNodes.push(Node{1.0, 0.0, 0.0, seq, kFrom_, -1, 0});  // (1, 0, 0) at :721
        pathExists_ = false;
        foundPaths_ = 0;
        const double oneMinusWeight = 1.0 - weight;
        while (!openNodes.empty()) {
            const Node node = openNodes.top();
            openNodes.pop();
            ++touchedNodes_;
            if (node.depth > kMaximumDepth) continue;  // :755
            if (node.vertex == kTo_) {                 // :766
                path_.clear();
                path_.push_back(node.vertex);
                for (int l = node.chain; l >= 0; l = chain[(size_t)l].prev) path_.push_back(chain[(size_t)l].vertex);
                std::reverse(path_.begin(), path_.end());
                SE3d pose;
                if (recoverPath(*kPoseGraph, path_, pose)) {
                    ++foundPaths_;
                    if (!test || test(pose)) poses_.push_back(pose);
                }
                break;  // kMaximumPathNumber = 1 (:799-800): the first recovered path ends the search
            }
            chain.push_back(Link{node.vertex, node.chain});
            const int here = (int)chain.size() - 1;
            if (std::find(nodeStates.begin(), nodeStates.end(), node.vertex) == nodeStates.end()) nodeStates.push_back(node.vertex);
            if (node.depth < kMaximumDepth)  // :817-820
                kPoseGraph->forEachEdgeOf(node.vertex, [&](const PoseGraphEdge& e) {
                    if (e.getScore() < kMinimumInlierRatio) return;  // :830
                    const ViewId next = node.vertex == e.getDestinationId() ? e.getSourceId() : e.getDestinationId();
                    const double edgeCost = std::min(node.edgeCost, e.getScore());                             // :843
                    const double nextCost = std::max(node.nextCost, kHeuristicsObject.getCost(next, kTo_));    // :847
                    const double combined = weight * edgeCost + oneMinusWeight * nextCost;                     // :851
                    if (std::find(nodeStates.begin(), nodeStates.end(), next) == nodeStates.end())             // :855
                        openNodes.push(Node{edgeCost, nextCost, combined, ++seq, next, here, node.depth + 1});
                });
        }
        pathExists_ = !poses_.empty();
    }

   protected:
    const PoseGraph* kPoseGraph;
    const _NodeCostHeuristics& kHeuristicsObject;
    const double weight;  // CostComparator::weight (graph_traversal.h:662, set at pose_graph_builder.h:828)
    const double kMinimumInlierRatio;
    const size_t kMaximumDepth;
};

}  // namespace reconstruction
