#!/usr/bin/env python3
"""Host-to-host throughput of pgi_estimate_pose_batch_host on BASELINE config 2 with page-locked buffers: K1 working on them
in place (default) against the copy pipeline (PGI_HOST_DIRECT=0); `ramp a,b,c ...` sweeps the pipeline's chunk sizes (pairs)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from pyposegraphbuilder import Engine, synthetic as S
    from pyposegraphbuilder import _lib as L
    P, N = 10000, 2000
    b = S.make_batch(np.arange(P), np.full(P, N))
    eng = Engine()
    px = [torch.from_numpy(np.ascontiguousarray(b[k], np.float32)).pin_memory().numpy() for k in ("x1", "y1", "x2", "y2")]
    pe = torch.zeros(P * L.EDGE_DTYPE.itemsize, dtype=torch.uint8).pin_memory().numpy().view(L.EDGE_DTYPE)
    pm = torch.zeros(P * N, dtype=torch.uint8).pin_memory().numpy()
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
    e, m = eng.estimate_pose_batch(db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.estimate_pose_batch(db)
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / 5
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        eng.estimate_pose_batch_host(*px, b["offsets"], 7.5e-4, seed=1, out=(pe, pm))
        ts.append(time.perf_counter() - t0)
    same = bool(np.array_equal(pm, m.cpu().numpy()) and np.array_equal(pe["E"], eng.edges_to_numpy(e)["E"]))
    best, med = min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]
    print("direct=%s chunk=%s: resident %.2f ms | host-to-host best %.2f ms median %.2f ms = %.3f M edges/s (median)  identical=%s" % (
        os.environ.get("PGI_HOST_DIRECT", "1"), os.environ.get("PGI_HOST_CHUNK_PAIRS", "-"), res * 1e3, best * 1e3, med * 1e3,
        P / med / 1e6, same))
    eng.close()
    sys.exit(0)
envs = [{"PGI_HOST_DIRECT": "0", "PGI_HOST_CHUNK_PAIRS": v} for v in sys.argv[2:]] if len(sys.argv) > 2 and sys.argv[1] == "ramp" else None
for env in envs or [{}, {"PGI_HOST_DIRECT": "0"}, {}, {"PGI_HOST_DIRECT": "0"}, {"PGI_HOST_DIRECT": "0", "PGI_HOST_CHUNKS": "2,2"}]:
    r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-2000:])
