"""Multi-GPU sharding of the pose-estimation path.

Image pairs are independent units: rank r of W owns a contiguous, sum(N)-balanced block of the pair
list, estimates its edges on its own GPU (no data-path collective), and the path's single exchange
step is one all-gather of the fixed-size edge records (200 B each; RCCL over xGMI when the backend is
"nccl") so that every rank holds the full edge table for the replicated rotation averaging.
torch.distributed is plumbing only.
"""
import numpy as np

from ._lib import EDGE_DTYPE


def shard_bounds(sizes, world):
    """Contiguous blocks [lo_r, hi_r) of the pair list with balanced total row counts."""
    sizes = np.asarray(sizes, np.int64)
    P = len(sizes)
    if world <= 1 or P == 0:
        return [(0, P)] + [(P, P)] * (max(world, 1) - 1)
    csum = np.concatenate([[0], np.cumsum(np.maximum(sizes, 1))])
    cuts = [0]
    for r in range(1, world):
        target = csum[-1] * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), P))
    cuts.append(P)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def allgather_edges(local_edges, counts, group=None):
    """local_edges: uint8 tensor [P_r, 200] on this rank's device (or CPU for gloo); counts: pairs per rank.

    Returns the [sum(counts), 200] table in rank order == global pair order (blocks are contiguous).
    Uneven shards are padded to the largest block for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = EDGE_DTYPE.itemsize
    pmax = int(max(counts)) if len(counts) else 0
    pad = torch.zeros((pmax, width), dtype=torch.uint8, device=local_edges.device)
    pad[:local_edges.shape[0]] = local_edges
    out = torch.empty((world * pmax, width), dtype=torch.uint8, device=local_edges.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * pmax:r * pmax + int(counts[r])] for r in range(world)]
    return torch.cat(parts, 0)


def edges_to_rotation_graph(edges_np, src, dst):
    """Edge records of the OK edges -> (src, dst, R_rel, weight) for rotation_average; weight = inlier ratio."""
    ok = edges_np["status"] == 1
    return (np.asarray(src)[ok], np.asarray(dst)[ok], edges_np["R"][ok].reshape(-1, 3, 3), np.ones(int(ok.sum())))
