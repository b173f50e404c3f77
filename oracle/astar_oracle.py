"""astar_oracle.py -- CPU ORACLE for the A* pose-guess search (test infrastructure, NOT product code).

Restates, in plain Python, the reference's
  AStarTraversal<ImageSimilarityHeuristics>::getPath   graph_traversal.h:679-870
  PoseGraphTraversal::recoverPath                       graph_traversal.h:290-348
  CostComparator (max-heap on the combined weight)      graph_traversal.h:656-677
as called by PoseGraphBuilder::findPath (pose_graph_builder.h:785-862): kReturnMultiplePaths = true,
kMinimumInlierRatio = 0.0, kMaximumDepth = kMaximumSearchDepth, kMaximumPathNumber = 1 (the constexpr at :799
shadows the member), so at most ONE path is ever recovered and tested.

PARITY UNPINNED against the reference binary (it cannot be built here).  One behaviour is made explicit
because the reference leaves it to the STL: heap ties on the combined weight are broken by insertion order
(earlier first).  The in-traversal pose test (graph_traversal.h:787-797) is a callback; the GPU build batches
those tests after the search, which is equivalent because the search stops at the first recovered path.
"""
import heapq

import numpy as np


class PoseGraph:
    """pose_graph.h:62-226 -- vertices, directed edges (src,dst) -> (R, t, score), edges per vertex in insertion order."""

    def __init__(self):
        self.vertices = set()
        self.edges = {}
        self.by_vertex = {}

    def add_vertex(self, v):
        self.vertices.add(v)

    def has_edge(self, s, d):
        return (s, d) in self.edges

    def add_edge(self, s, d, R, t, score):
        if s not in self.vertices or d not in self.vertices or (s, d) in self.edges:
            return False
        self.edges[(s, d)] = (np.asarray(R, float).reshape(3, 3), np.asarray(t, float), float(score))
        self.by_vertex.setdefault(s, []).append((s, d))
        self.by_vertex.setdefault(d, []).append((s, d))
        return True


def recover_path(graph, path):
    """graph_traversal.h:290-348: pose <- T_edge * pose, or T_edge^-1 * pose when the edge is stored reversed."""
    R, t = np.eye(3), np.zeros(3)
    for a, b in zip(path[:-1], path[1:]):
        if graph.has_edge(a, b):
            Re, te, _ = graph.edges[(a, b)]
        elif graph.has_edge(b, a):
            Rs, ts, _ = graph.edges[(b, a)]
            Re, te = Rs.T, -Rs.T @ ts
        else:
            return None
        R, t = Re @ R, Re @ t + te
    return R, t


def astar_get_path(graph, similarity, src, dst, weight=0.8, max_depth=5, min_inlier_ratio=0.0):
    """Returns (path or None, (R,t) or None, touched_nodes).  similarity(a, b) -> [0,1] clamp applied here."""
    heap = []  # entries (-combined, seq, edge_cost, next_cost, vertex, parents, depth); max combined first
    seq = 0
    heapq.heappush(heap, (-0.0, seq, 1.0, 0.0, src, (), 0))   # (1, 0, 0) at graph_traversal.h:721
    state = {}
    touched = 0
    while heap:
        negw, _, ecost, ncost, v, parents, depth = heapq.heappop(heap)
        touched += 1
        if depth > max_depth:                                   # :755
            continue
        if v == dst:                                            # :766
            path = list(parents) + [v]
            pose = recover_path(graph, path)                    # :774
            return path, pose, touched                          # first recovered path ends the search (:799-800)
        parents2 = parents + (v,)
        state[v] = "open"
        if depth < max_depth:                                   # :820
            for (s, d) in graph.by_vertex.get(v, []):
                _, _, score = graph.edges[(s, d)]
                if score < min_inlier_ratio:
                    continue
                nxt = s if v == d else d                         # :838-840
                edge_cost = min(ecost, score)                    # :843-844
                next_cost = max(ncost, min(max(similarity(nxt, dst), 0.0), 1.0))   # :847-848, clamp :594
                comb = weight * edge_cost + (1.0 - weight) * next_cost             # :851-852
                if nxt not in state:                             # :855-856 (no 'Seen' marking: multi-path mode)
                    seq += 1
                    heapq.heappush(heap, (-comb, seq, edge_cost, next_cost, nxt, parents2, depth + 1))
        state[v] = "closed"
    return None, None, touched


class UnionFind:
    """Replacement for visibility_table.h (whose transitive closure is buggy)."""

    def __init__(self, n):
        self.p = list(range(n))

    def find(self, a):
        while self.p[a] != a:
            self.p[a] = self.p[self.p[a]]
            a = self.p[a]
        return a

    def add_link(self, a, b):
        a, b = self.find(a), self.find(b)
        if a != b:
            self.p[max(a, b)] = min(a, b)

    def has_link(self, a, b):
        return self.find(a) == self.find(b)
