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


class Communicator:
    """Installs the C-ABI exchange (pgi_comm_*, include/pgi.h) in an Engine, bootstrapped by torch.distributed.

    transport "rccl": rank 0 creates the RCCL unique id inside libpgi.so, torch.distributed only ships its 128 bytes;
    the all-gather of the edge records then runs inside libpgi.so on the engine's stream (RCCL over xGMI).
    transport "host": an all-gather-v over host memory through the process group (gloo) -- for ranks that share one
    device (RCCL rejects duplicate devices; the one-GPU test box) and for CPU-side tests.
    "auto" picks RCCL iff every rank reports a different device."""

    def __init__(self, engine, group=None, transport="auto"):
        import ctypes as C
        import socket
        import torch
        import torch.distributed as dist
        from . import _lib as L
        self.engine, self.group = engine, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.transport = "none"
        self.rccl_version = 0
        if self.world == 1:
            return
        if transport == "auto":
            props = torch.cuda.get_device_properties(engine.device)
            ident = (socket.gethostname(), getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", engine.device.index),
                     getattr(props, "pci_device_id", 0), str(getattr(props, "uuid", "")))
            seen = [None] * self.world
            dist.all_gather_object(seen, ident, group=group)
            transport = "rccl" if len(set(seen)) == self.world else "host"
        lib = L.load()
        if transport == "rccl":
            # Every step that can fail on one rank only is followed by an agreement round, so that the ranks either all
            # hold a communicator or all raise (a rank that raises alone would leave the others blocked in a collective).
            ident, err = [None], None
            ver = C.c_int(0)
            usable = lib.pgi_comm_rccl_probe(C.byref(ver)) == 0  # librccl opens and has every entry point (no collective yet)
            probes = [None] * self.world
            dist.all_gather_object(probes, (usable, "" if usable else L.last_error()), group=group)
            if not all(p[0] for p in probes):
                raise L.PgiError("RCCL is not usable on rank(s) %s" % {r: p[1] for r, p in enumerate(probes) if not p[0]})
            self.rccl_version = int(ver.value)
            if self.rank == 0:
                buf = (C.c_uint8 * L.COMM_ID_BYTES)()
                rc = lib.pgi_comm_unique_id(buf)
                ident = [bytes(buf) if rc == 0 else None]
            dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if ident[0] is None:
                raise L.PgiError("rank 0 could not create the RCCL unique id: " + (L.last_error() if self.rank == 0 else "see rank 0"))
            buf = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(ident[0])
            rc = lib.pgi_comm_init_rccl(engine._ctx, self.world, self.rank, buf)
            if rc != 0:
                err = L.last_error()
            oks = [None] * self.world
            dist.all_gather_object(oks, rc == 0, group=group)
            if not all(oks):
                if rc == 0:
                    lib.pgi_comm_destroy(engine._ctx)
                raise L.PgiError("pgi_comm_init_rccl failed on rank(s) %s%s" % ([r for r, o in enumerate(oks) if not o],
                                                                              ": " + err if err else ""))
        elif transport == "host":
            if dist.get_backend(group) != "gloo":
                raise L.PgiError("host transport needs a gloo process group (ranks sharing a device cannot use RCCL)")

            def gather(user, send, send_bytes, recv, recv_bytes, world):
                try:
                    sizes = [int(recv_bytes[r]) for r in range(world)]
                    pmax = max(sizes + [1])
                    mine = torch.zeros(pmax, dtype=torch.uint8)
                    if send_bytes:
                        mine[:send_bytes] = torch.frombuffer((C.c_uint8 * send_bytes).from_address(send), dtype=torch.uint8)
                    parts = [torch.empty(pmax, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, mine, group=self.group)
                    pos = 0
                    for r in range(world):
                        if sizes[r]:
                            C.memmove(recv + pos, parts[r].numpy().ctypes.data, sizes[r])
                        pos += sizes[r]
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    return 1
            self._callback = L.ALLGATHERV_FN(gather)  # keep alive as long as the communicator
            L.check(lib.pgi_comm_init_host(engine._ctx, self.world, self.rank, self._callback, None))
        else:
            raise ValueError("transport must be auto, rccl or host")
        self.transport = transport

    def allgather_edges(self, local_edges, counts, out=None):
        """Edge records of every rank, in rank order == global pair order for contiguous blocks."""
        return self.engine.allgather_edges(local_edges, counts, out)

    def close(self):
        if self.transport != "none" and getattr(self.engine, "_ctx", None):
            from . import _lib as L
            L.load().pgi_comm_destroy(self.engine._ctx)
        self.transport = "none"


def allgather_edges(local_edges, counts, group=None):
    """Plain torch.distributed variant (no engine): local_edges uint8 [P_r, 200] (CPU for gloo, device for nccl);
    counts: pairs per rank.  Returns the [sum(counts), 200] table in rank order.  Uneven shards are padded to the
    largest block for the collective.  The product path uses Communicator / pgi_allgather_edges instead."""
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


def edges_to_rotation_graph(edges_np, src, dst, rows=None):
    """Edge records of the OK edges -> (src, dst, R_rel, weight) for rotation_average.  weight = inlier ratio
    n_inl / rows (the reference's edge score, pose_graph_builder.h:645-654); uniform weights when rows is None."""
    ok = edges_np["status"] == 1
    if rows is None:
        w = np.ones(int(ok.sum()))
    else:
        w = edges_np["n_inl"][ok].astype(np.float64) / np.maximum(np.asarray(rows, np.float64)[ok], 1.0)
    return np.asarray(src)[ok], np.asarray(dst)[ok], edges_np["R"][ok].reshape(-1, 3, 3), w
