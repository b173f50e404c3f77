"""Re-derives tests/golden/golden_v1_nister_lo.npz after round 6 re-defined the record's E (run in the build container:
`python tests/golden/make_golden_nister_lo.py`).

The file pins the estimator's behaviour BEFORE the hybrid linear / n-point refit became the default (its first version was
taken from the repository history, commit b743299^).  Round 6 changed what pgi_edge.E / pgo_edge.E holds -- the essential
matrix of the RETURNED POSE, [t]x R at unit norm, instead of the fitted f32 model -- so the E columns of the pin move by f32
rounding (< 1e-6) and nothing else may: this script recomputes the three outputs with lo_linear_pct = 0, REFUSES to write
unless every other field and every mask equals the file it replaces byte for byte and E moved by less than 1e-6, and only
then writes the new E."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "pose-graph-initialization_amd"))
import oracle_lib as O  # noqa: E402


def main():
    G = np.load(os.path.join(HERE, "golden_v1.npz"))
    path = os.path.join(HERE, "golden_v1_nister_lo.npz")
    old = np.load(path)
    new = {}
    for tag, kw, guess in (("", {}, False), ("_fixed", {"fixed_budget": 96}, False), ("_guess", {}, True)):
        out, masks = O.estimate_pose_batch(G["ep_x1"], G["ep_y1"], G["ep_x2"], G["ep_y2"], G["ep_offsets"], G["ep_thr"],
                                           O.default_params(lo_linear_pct=0, **kw), int(G["ep_seed"]), pair_id_base=9000,
                                           guesses=G["ep_guesses"] if guess else None,
                                           has_guess=np.ones(8, np.uint8) if guess else None)
        was = old["ep_out" + tag]
        for f in was.dtype.names:
            if f == "E":
                d = np.minimum(np.abs(out["E"] - was["E"]).max(1), np.abs(out["E"] + was["E"]).max(1))
                assert d.max() < 1e-6, ("E moved by more than f32 rounding", tag, d.max())
            else:
                assert np.array_equal(out[f], was[f]), (tag, f)
        assert np.array_equal(masks, old["ep_masks" + tag]), tag
        new["ep_out" + tag], new["ep_masks" + tag] = out, masks
    for k in old.files:
        if k not in new:
            new[k] = old[k]
    np.savez_compressed(path, **new)
    print("wrote", path, "(every field but E byte-identical to the file it replaces)")


if __name__ == "__main__":
    main()
