import sys, os, json, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pose-graph-initialization_amd"); sys.path.insert(0, "/root/repo/tests")
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", "/root/repo/bench.py"); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
mode = sys.argv[1]
eng = torch = None
if mode != "bare":
    import torch
    from pyposegraphbuilder import Engine
    eng = Engine()
    if mode == "busy":   # what bench.py has done before the graph legs: a resident batch and a few launches
        from pyposegraphbuilder import synthetic as S
        import numpy as np
        b = S.make_batch(np.arange(10000), 2000)
        db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
        for _ in range(5): eng.estimate_pose_batch(db)
        torch.cuda.synchronize()
g = B.graph_level(1, eng=None, torch=None)
v = g["v5000"]
print(mode, {k: (v[k]["seconds"], v[k].get("all_repetitions_s")) for k in v if isinstance(v[k], dict) and "seconds" in v[k]})
