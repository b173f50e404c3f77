"""One rank of the Python multi-GPU path (launched by tests/test_distributed_gpu.py with RANK / WORLD_SIZE / MASTER_*).

shard -> estimate on this rank's block -> Communicator.allgather_edges (pgi_allgather_edges in the C ABI) ->
pgi_rotation_average_edges on the gathered device table.  Writes <prefix>.<rank> = edge table + rotations."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))


def main():
    import torch
    import torch.distributed as dist
    from pyposegraphbuilder import Engine, synthetic as S
    from pyposegraphbuilder import distributed as D
    prefix = sys.argv[1]
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    rccl = os.environ.get("PGI_TEST_RCCL") == "1"  # one device per rank: the records travel by RCCL inside libpgi.so
    torch.cuda.set_device(rank if rccl else 0)     # default: both ranks share the one visible GPU (host transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V = 120
    g = S.make_scene_graph(V, k=6, seed=4, median_corr=300, max_corr=1500, outlier_pair_frac=0.05)
    b, pairs = g["batch"], g["pairs"]
    sizes = np.diff(b["offsets"].astype(np.int64))
    bounds = D.shard_bounds(sizes, world)
    lo, hi = bounds[rank]
    eng = Engine()
    comm = D.Communicator(eng, transport="rccl" if rccl and world > 1 else "auto")
    if rccl and world > 1:
        assert eng.comm_info() == (world, rank, "rccl")
    r0, r1 = int(b["offsets"][lo]), int(b["offsets"][hi])
    db = eng.upload(b["x1"][r0:r1], b["y1"][r0:r1], b["x2"][r0:r1], b["y2"][r0:r1],
                    b["offsets"][lo:hi + 1] - b["offsets"][lo], 7.5e-4, seed=9, pair_id_base=lo)
    edges, _ = eng.estimate_pose_batch(db)
    table = comm.allgather_edges(edges, [h - l for l, h in bounds])
    R, iters, used = eng.rotation_average_edges(table, pairs[:, 0], pairs[:, 1], sizes, V)
    torch.cuda.synchronize()
    got = eng.edges_to_numpy(table)
    assert len(got) == len(pairs) and used == int((got["status"] == 1).sum())
    # the host-table route (pgi_rotation_average) gives the same rotations as the device-table route
    src, dst, Rrel, w = D.edges_to_rotation_graph(got, pairs[:, 0], pairs[:, 1], sizes)
    R2, iters2 = eng.rotation_average(src, dst, Rrel, w, V)
    assert iters2 == iters and np.array_equal(R, R2)
    with open("%s.%d" % (prefix, rank), "wb") as f:
        f.write(table.cpu().numpy().tobytes())
        f.write(np.ascontiguousarray(R).tobytes())
    print("rank %d/%d transport=%s edges_ok=%d iters=%d" % (rank, world, comm.transport if world > 1 else "none", used, iters))
    comm.close()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
