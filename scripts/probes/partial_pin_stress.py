#!/usr/bin/env python3
"""Stress of the host-pointer call around partly page-locked buffers (tests/test_gpu_parity.py::
test_partially_page_locked_buffers_are_refused): arrays from the brk heap (glibc's mmap threshold raised first, as an earlier
test in a long pytest process does), page-locked in part, refused, unlocked, then used pageable.  Usage: partial_pin_stress.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder import Engine, _lib as L, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng = Engine()
for sz in (8, 16, 32, 48):   # freeing mmapped blocks raises the threshold: later arrays of this size live in the main heap
    a = np.zeros(sz << 20, np.uint8); a[::4096] = 1; del a
P = 700
b = S.make_batch(np.arange(9100, 9100 + P), 600)
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=5, pair_id_base=9)
e, m = eng.estimate_pose_batch(db)
ref, ref_m = eng.edges_to_numpy(e), m.cpu().numpy()
for it in range(n):
    xs = [np.array(b[k], np.float32) for k in ("x1", "y1", "x2", "y2")]
    out = (np.zeros(P, ref.dtype), np.zeros(len(ref_m), np.uint8))
    if it == 0:
        print("array addresses: %s" % " ".join(hex(a.ctypes.data) for a in xs + list(out)), flush=True)
    if it % 3 == 0:   # as the test before it does: everything page-locked, worked on in place
        eng.pin(*xs, *out)
        try:
            got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=5, pair_id_base=9, out=out)
        finally:
            eng.unpin(*xs, *out)
        assert np.array_equal(got_m, ref_m) and np.array_equal(got["E"], ref["E"])
    for part in ([xs[0][:len(xs[0]) // 2]], [out[1][:len(out[1]) // 3]], [xs[2][len(xs[2]) // 2:]]):
        eng.pin(*part)
        try:
            try:
                eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=5, pair_id_base=9, out=out)
                raise SystemExit("not refused at iteration %d" % it)
            except L.PgiError as ex:
                assert "only in part" in str(ex), ex
        finally:
            eng.unpin(*part)
    got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=5, pair_id_base=9, out=out)
    assert np.array_equal(got_m, ref_m) and np.array_equal(got["E"], ref["E"]), it
    print(".", end="", flush=True)
print("\n%d iterations without a fault" % n)
