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
    // Sparse table (round 4): only the listed pairs are stored -- a hash map keyed by the unordered pair -- and every other
    // off-diagonal cell reads `fill`, the diagonal 1.0 (what a dense table holds after setting every i < j cell to `fill`).
    // The dense N x N form above is what the reference allocates (imagesimilarity_graph.h:55-65) and what loadFromFile
    // fills; callers that hand over a candidate LIST (pgih_run_pairs, the scene drivers) need a few values per view, and a
    // dense table for them costs 200 MB and 12.5 M stores at 5000 views, 3.2 GB at 20 000.
    struct Sparse {};
    SimilarityTable(Sparse, size_t num_imgs, double fill_, double image_similarity_threshold_ = 0.0, bool build_priority_queue_ = false)
        : size(num_imgs),
          build_priority_queue(build_priority_queue_),
          image_similarity_threshold(image_similarity_threshold_),
          sparse(true),
          fill(fill_) {}
    bool isSparse() const { return sparse; }
    bool setSimilarity(ViewId from, ViewId to, double val) {
        if (from >= size || to >= size) return false;
        if (build_priority_queue && from != to && image_similarity_threshold <= val) {
            views.insert(from);
            views.insert(to);
            view_pair_queue.emplace(val, from, to);
        }
        if (sparse) {
            if (from == to) return true;
            auto ins = listed.emplace(key(from, to), val);
            if (ins.second) {  // a new pair: it also joins the two views' rows
                if (rows.size() < size) rows.resize(size);
                rows[from].emplace_back((uint32_t)to, val);
                rows[to].emplace_back((uint32_t)from, val);
            } else {
                ins.first->second = val;
                for (auto& e : rows[from]) if (e.first == (uint32_t)to) e.second = val;
                for (auto& e : rows[to]) if (e.first == (uint32_t)from) e.second = val;
            }
            return true;
        }
        similarity[from][to] = val;
        similarity[to][from] = val;
        return true;
    }
    double getSimilarity(ViewId from, ViewId to) const {
        if (from >= size || to >= size) return -1.0;  // NO_SUCH_VERTEX
        if (sparse) {
            if (from == to) return 1.0;
            auto it = listed.find(key(from, to));
            return it == listed.end() ? fill : it->second;
        }
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
        if (sparse) return false;  // the file format is the dense matrix
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
        // a file is written cell by cell and need not be symmetric; rowTo() may hand out a row in place of a column only
        // while it is (setSimilarity keeps it so; nothing else writes cells)
        symmetric = true;
        for (size_t i = 0; i < N && symmetric; ++i)
            for (size_t j = i + 1; j < N; ++j)
                if (similarity[i][j] != similarity[j][i]) { symmetric = false; break; }
        return true;
    }
    // All similarities TO one view, for a traversal that asks getSimilarity(next, to) for many `next` and one `to` (A*'s
    // heuristic, graph_traversal.h:847): the dense table's row, or -- sparse -- the view's listed values scattered into the
    // calling thread's stamped array, so that each question is one load instead of one hash lookup.
    class RowTo {
       public:
        double operator()(ViewId from) const {
            if (dense) return from < n ? dense[from] : -1.0;
            if (from >= n) return -1.0;  // NO_SUCH_VERTEX
            if (from == to) return 1.0;
            return stamp[from] == epoch ? value[from] : fill;
        }

       private:
        friend class SimilarityTable;
        const double* dense = nullptr;
        const double* value = nullptr;   // the calling thread's scratch (sized to the table: never resized while in use)
        const uint32_t* stamp = nullptr;
        uint32_t epoch = 0;
        size_t n = 0;
        ViewId to = 0;
        double fill = 0.0;
    };
    RowTo rowTo(ViewId to) const {
        RowTo r;
        r.n = size;
        r.to = to;
        r.fill = fill;
        struct Scratch { std::vector<double> value; std::vector<uint32_t> stamp; uint32_t epoch = 0; };
        static thread_local Scratch sc;
        if (!sparse) {
            if (to >= size) { r.n = 0; return r; }
            if (symmetric) { r.dense = similarity[to].data(); return r; }  // row == column: no gather
            // an asymmetric file (loadFromFile): the heuristic is similarity[next][to] (graph_traversal.h:847), i.e. the
            // COLUMN of `to` -- gathered once per search into the calling thread's scratch
            if (sc.value.size() < size) { sc.stamp.assign(size, 0u); sc.value.assign(size, 0.0); sc.epoch = 0; }
            for (size_t from = 0; from < size; ++from) sc.value[from] = similarity[from][to];
            r.dense = sc.value.data();
            return r;
        }
        if (sc.stamp.size() < size) { sc.stamp.assign(size, 0u); sc.value.assign(size, 0.0); sc.epoch = 0; }
        if (++sc.epoch == 0) { std::fill(sc.stamp.begin(), sc.stamp.end(), 0u); sc.epoch = 1; }
        if (to < rows.size())
            for (const auto& e : rows[to]) { sc.value[e.first] = e.second; sc.stamp[e.first] = sc.epoch; }
        r.value = sc.value.data();
        r.stamp = sc.stamp.data();
        r.epoch = sc.epoch;
        return r;
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
    bool sparse = false;
    bool symmetric = true;  // dense mode: similarity[i][j] == similarity[j][i] for all cells (false only after an asymmetric file)
    double fill = 0.0;
    static uint64_t key(ViewId a, ViewId b) { return a < b ? ((uint64_t)a << 32) | (uint64_t)b : ((uint64_t)b << 32) | (uint64_t)a; }
    std::unordered_map<uint64_t, double> listed;  // sparse mode: unordered pair -> similarity (ids < 2^32)
    std::vector<std::vector<std::pair<uint32_t, double>>> rows;  // sparse mode: per view, its listed (other view, similarity)
};

class VisibilityTable {  // "is there already a path between two views?" (pose_graph_builder.h:456-457, 692)
   public:
    explicit VisibilityTable(size_t n) : parent(n) {
        for (size_t i = 0; i < n; ++i) parent[i] = i;
    }
    bool hasLink(ViewId a, ViewId b) const { return find(a) == find(b); }
    void addLink(ViewId a, ViewId b) {  // (single writer; hasLink may run concurrently only between commits)
        const ViewId ra = find(a), rb = find(b), root = std::min(ra, rb);
        if (ra != rb) parent[std::max(ra, rb)] = root;
        for (ViewId v : {a, b})  // both walks now end at `root`: point them at it, so the chains hasLink follows stay short
            while (parent[v] != root && v != root) {
                const ViewId next = parent[v];
                parent[v] = root;
                v = next;
            }
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
    // getCost(., kTo_) for one search: the table's row towards the target, looked up once (SimilarityTable::rowTo)
    struct CostsTo {
        SimilarityTable::RowTo row;
        double operator()(const ViewId& a) const { return std::clamp(row(a), 0.0, 1.0); }
    };
    CostsTo costsTo(const ViewId& kTo_) const { return CostsTo{kSimilarityTable.rowTo(kTo_)}; }

   protected:
    const SimilarityTable& kSimilarityTable;
};

// graph_traversal.h:290-348: pose <- T_edge * pose, or T_edge^-1 * pose for a reversed edge
inline bool recoverPath(const PoseGraph& g, const std::vector<ViewId>& path, SE3d& pose_, bool frozen = false) {
    pose_ = SE3d();
    for (size_t i = 1; i < path.size(); ++i) {
        const ViewId a = path[i - 1], b = path[i];
        if (frozen) {  // no writer runs: plain lookups, no copies (PoseGraph::findEdgeFrozen)
            if (const PoseGraphEdge* e = g.findEdgeFrozen({a, b})) pose_ = e->getValue().getTransform() * pose_;
            else if (const PoseGraphEdge* r = g.findEdgeFrozen({b, a})) pose_ = r->getValue().getTransform().inverse() * pose_;
            else return false;
            continue;
        }
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

    // The caller promises that nothing modifies the pose graph while searches run (the wave scheduler: between two
    // commits); the traversal then reads it without the shared lock, so kCoreNumber search threads do not contend.
    void setGraphFrozen(bool on) { frozen = on; }

    // graph_traversal.h:679-870.  path_ / poses_ receive the (single) recovered path and its chained pose;
    // with a pose test the pose is returned only if the test accepts it (:787-797).
    void getPath(const ViewId kFrom_, const ViewId kTo_, std::vector<ViewId>& path_, std::vector<SE3d>& poses_,
                 size_t& touchedNodes_, size_t& foundPaths_, bool& pathExists_, const PoseTest& test = PoseTest()) const {
        // (the reference copies the parent list into every open node; here an open node points at a link of a shared chain.
        //  The search's working memory -- heap, chain, the set of expanded vertices -- belongs to the calling thread and is
        //  reused from search to search: no allocation per search once warm, and "already expanded?" is one load: a stamp
        //  per view id for ids below 2^22, a hash set above)
        struct Link { ViewId vertex; int prev; };
        struct Node {  // an open node's payload; the heap itself holds 16-byte keys that point here
            double edgeCost, nextCost;
            ViewId vertex;
            int chain;  // link of the parent vertex, -1 at the start vertex
            uint32_t depth;
        };
        struct Key { double combined; uint32_t seq, node; };
        auto worse = [](const Key& a, const Key& b) {  // max-heap on combined, earlier insertion first
            if (a.combined != b.combined) return a.combined < b.combined;
            return a.seq > b.seq;
        };
        struct Scratch {
            std::vector<Key> heap;
            std::vector<Node> nodes;
            std::vector<Link> chain;
            std::vector<uint32_t> stamp;
            uint32_t epoch = 0;
            std::unordered_set<ViewId> big;  // ids >= kStampLimit
        };
        static thread_local Scratch tls;
        Scratch& sc = tls;  // (one thread-local lookup per search, not one per visited edge)
        constexpr ViewId kStampLimit = (ViewId)1 << 22;
        sc.heap.clear();
        sc.nodes.clear();
        sc.chain.clear();
        sc.big.clear();
        if (++sc.epoch == 0) {  // wrapped: forget every stamp
            std::fill(sc.stamp.begin(), sc.stamp.end(), 0u);
            sc.epoch = 1;
        }
        const uint32_t epoch = sc.epoch;
        auto expanded = [&](ViewId v) {
            if (v < kStampLimit) return v < sc.stamp.size() && sc.stamp[v] == epoch;
            return sc.big.count(v) != 0;
        };
        auto markExpanded = [&](ViewId v) {
            if (v < kStampLimit) {
                if (v >= sc.stamp.size()) sc.stamp.resize(std::max<size_t>(v + 1, 2 * sc.stamp.size()), 0u);
                sc.stamp[v] = epoch;
            } else {
                sc.big.insert(v);
            }
        };
        std::vector<Key>& openNodes = sc.heap;
        std::vector<Node>& nodes = sc.nodes;
        std::vector<Link>& chain = sc.chain;
        uint32_t seq = 0;
        nodes.push_back(Node{1.0, 0.0, kFrom_, -1, 0});  // (1, 0, 0) at :721
        openNodes.push_back(Key{0.0, seq, 0});
        pathExists_ = false;
        foundPaths_ = 0;
        const double oneMinusWeight = 1.0 - weight;
        const auto costTo = kHeuristicsObject.costsTo(kTo_);  // == getCost(., kTo_)
        while (!openNodes.empty()) {
            std::pop_heap(openNodes.begin(), openNodes.end(), worse);
            const Node node = nodes[openNodes.back().node];
            openNodes.pop_back();
            ++touchedNodes_;
            if (node.depth > kMaximumDepth) continue;  // :755
            if (node.vertex == kTo_) {                 // :766
                path_.clear();
                path_.push_back(node.vertex);
                for (int l = node.chain; l >= 0; l = chain[(size_t)l].prev) path_.push_back(chain[(size_t)l].vertex);
                std::reverse(path_.begin(), path_.end());
                SE3d pose;
                if (recoverPath(*kPoseGraph, path_, pose, frozen)) {
                    ++foundPaths_;
                    if (!test || test(pose)) poses_.push_back(pose);
                }
                break;  // kMaximumPathNumber = 1 (:799-800): the first recovered path ends the search
            }
            chain.push_back(Link{node.vertex, node.chain});
            const int here = (int)chain.size() - 1;
            markExpanded(node.vertex);
            if (node.depth < kMaximumDepth) {  // :817-820
                auto visitNeighbour = [&](const ViewId next, const double score) {
                    if (score < kMinimumInlierRatio) return;  // :830
                    if (expanded(next)) return;                                                    // :855
                    const double edgeCost = std::min(node.edgeCost, score);                        // :843
                    const double nextCost = std::max(node.nextCost, costTo(next));                 // :847
                    const double combined = weight * edgeCost + oneMinusWeight * nextCost;         // :851
                    nodes.push_back(Node{edgeCost, nextCost, next, here, node.depth + 1});
                    openNodes.push_back(Key{combined, ++seq, (uint32_t)nodes.size() - 1});
                    std::push_heap(openNodes.begin(), openNodes.end(), worse);
                };
                if (frozen)
                    kPoseGraph->forEachNeighbourFrozen(node.vertex, visitNeighbour);
                else
                    kPoseGraph->forEachEdgeOf(node.vertex, [&](const PoseGraphEdge& e) {
                        visitNeighbour(node.vertex == e.getDestinationId() ? e.getSourceId() : e.getDestinationId(), e.getScore());
                    });
            }
        }
        pathExists_ = !poses_.empty();
    }

   protected:
    const PoseGraph* kPoseGraph;
    const _NodeCostHeuristics& kHeuristicsObject;
    const double weight;  // CostComparator::weight (graph_traversal.h:662, set at pose_graph_builder.h:828)
    const double kMinimumInlierRatio;
    const size_t kMaximumDepth;
    bool frozen = false;
};

}  // namespace reconstruction
