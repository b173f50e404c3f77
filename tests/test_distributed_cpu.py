"""world_size-2 gloo tests (CPU) of the N>1 path: sharding + the all-gather of edge records."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyposegraphbuilder import distributed as D
from pyposegraphbuilder._lib import EDGE_DTYPE


def test_shard_bounds_balanced_contiguous():
    rng = np.random.default_rng(0)
    sizes = rng.integers(50, 4000, 1000)
    for world in (1, 2, 3, 4, 8):
        b = D.shard_bounds(sizes, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == 1000
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        loads = np.array([sizes[lo:hi].sum() for lo, hi in b])
        assert loads.max() - loads.min() <= 4000 * 2  # within a couple of pairs of each other
    assert D.shard_bounds([], 4) == [(0, 0)] * 4
    assert D.shard_bounds([10, 10], 4)[0] == (0, 0) or sum(hi - lo for lo, hi in D.shard_bounds([10, 10], 4)) == 2


def _worker(rank, world, port, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = D.shard_bounds(sizes, world)
    lo, hi = bounds[rank]
    # stand-in for the GPU stage: a deterministic record per global pair id
    rec = np.zeros(hi - lo, EDGE_DTYPE)
    rec["n_inl"] = np.arange(lo, hi)
    rec["status"] = 1
    rec["R"] = np.arange(lo, hi)[:, None] + np.arange(9)[None, :]
    local = torch.from_numpy(rec.view(np.uint8).reshape(hi - lo, EDGE_DTYPE.itemsize).copy())
    full = D.allgather_edges(local, [h - l for l, h in bounds])
    got = full.numpy().view(EDGE_DTYPE).reshape(-1)
    ok = (len(got) == len(sizes) and np.array_equal(got["n_inl"], np.arange(len(sizes)))
          and np.array_equal(got["R"][:, 0], np.arange(len(sizes), dtype=float)))
    q.put((rank, bool(ok), full.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_edges_gloo_world2():
    port = _free_port()
    sizes = [100, 3000, 50, 50, 700, 2000, 64, 900, 1200]  # uneven shards (5 vs 4 pairs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]  # every rank holds the identical global edge table


# ---- the C++ side of the multi-rank path (host/distributed.hpp): no torch, no GPU ------------------------------
import subprocess  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASTAR_EXE = os.path.join(ROOT, "pose-graph-initialization_amd", "test_astar")


def _free_port():
    """A rendezvous port BELOW the kernel's ephemeral range (32768+ on Linux): the ranks' clients retry connect() until rank 0
    listens, and a port inside that range can be handed to one of them as its own source port in the meantime."""
    import random
    for _ in range(200):
        port = random.randint(20000, 29999)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        return port
    raise RuntimeError("no free port in 20000-29999")


def test_cpp_shard_bounds_equal_python(tmp_path):
    rng = np.random.default_rng(3)
    for case, sizes in enumerate([rng.integers(50, 4000, 777), [0, 0, 5, 0], [10, 10], [7], rng.integers(1, 3, 40)]):
        f = tmp_path / ("sizes%d.txt" % case)
        f.write_text(" ".join(str(int(x)) for x in sizes))
        for world in (1, 2, 3, 8):
            out = tmp_path / "b.txt"
            subprocess.run([ASTAR_EXE, "shards", str(f), str(world), str(out)], check=True, timeout=60)
            got = [tuple(int(v) for v in line.split()) for line in open(out)]
            assert got == [(int(a), int(b)) for a, b in D.shard_bounds(sizes, world)], (case, world)


def _cpp_hostcomm(tmp_path, world):
    port = _free_port()
    prefix = str(tmp_path / "hc")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([ASTAR_EXE, "hostcomm", prefix], env=env, stderr=subprocess.PIPE))
    for p in procs:
        assert p.wait(timeout=120) == 0, p.stderr.read()
    blobs = [open("%s.%d" % (prefix, r), "rb").read() for r in range(world)]
    assert all(bl == blobs[0] for bl in blobs)
    b = np.frombuffer(blobs[0], np.uint8)
    assert np.array_equal(b[:128], (np.arange(128) * 7 + 3).astype(np.uint8))
    sizes = [0 if r == 1 else 1000 * (r + 1) + 13 for r in range(world)]   # tests/cpp/test_astar.cpp hostcomm: rank 1 sends nothing
    pos = 128
    for r, n in enumerate(sizes):
        assert np.array_equal(b[pos:pos + n], (r * 31 + np.arange(n)).astype(np.uint8))
        pos += n
    recs = np.frombuffer(blobs[0][pos:], np.uint32).reshape(world, 2)
    assert np.array_equal(recs, [[r, r * r] for r in range(world)])


def test_cpp_hostcomm_world3(tmp_path):
    """The TCP star that ships the RCCL unique id and carries the host-side exchanges: broadcast, uneven all-gather
    (one empty block), fixed-size all-gather, barrier -- three processes, identical results on every rank."""
    _cpp_hostcomm(tmp_path, 3)


def test_cpp_hostcomm_world8(tmp_path):
    """The same protocol at the width BASELINE's configs 4/5 name (8 ranks of one node)."""
    _cpp_hostcomm(tmp_path, 8)
