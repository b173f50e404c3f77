"""oracle/pipeline_oracle.py on the CPU: the restated feature-level run recovers the scene it is given (the oracle is the
checker of tests/test_feature_pipeline.py::test_feature_pipeline_equals_the_cpu_restatement; here it is checked itself
against ground truth and against the properties the loop must have whatever the mode)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib as O  # noqa: E402
import pipeline_oracle as PO  # noqa: E402
from pyposegraphbuilder import synthetic as S  # noqa: E402
from test_feature_pipeline import small_scene  # noqa: E402


@pytest.mark.parametrize("kw", [dict(path_finding=False, hashing=False), dict(hashing=False), dict(), dict(rotation_guided=True)],
                         ids=["plain", "astar", "hashing", "hashing_guided"])
def test_restated_feature_run_recovers_the_scene(kw):
    views, poses, cam, sim, pairs = small_scene()
    lookup = lambda p, q: 1.0 if p == q else float(sim[p, q])
    trace = []
    st, edges, fragile = PO.run_features(O, views, cam, pairs, lookup, 4, trace=trace, **kw)
    assert st["pairs_processed"] == len(pairs) and st["waves"] == 7 and st["edges_added"] == st["graph_edges"] == len(edges)
    assert st["matching_runs"] + st["quick_matching_runs"] == len(pairs)
    err = np.array([S.rot_err_deg(R, poses[d][0] @ poses[s][0].T) for (s, d), (sc, R, t) in edges.items()])
    assert np.median(err) < 0.1 and np.mean(err < 1.0) > 0.9
    for sc, R, t in edges.values():
        assert 0 < sc <= 1 and abs(np.linalg.det(R) - 1) < 1e-9 and abs(np.linalg.norm(t) - 1) < 1e-9
    if not kw.get("path_finding", True):
        assert st["paths_searched"] == st["poses_from_guess"] == 0
    else:   # the first wave has no graph to search; later waves do
        assert 0 < st["paths_found"] <= st["paths_searched"] < len(pairs) and st["poses_from_guess"] > 0
    if not kw.get("hashing", True):
        assert st["quick_matching_runs"] == st["guided_matching_runs"] == st["track_number"] == 0 and st["too_few_matches"] == 0
        assert len(edges) == len(pairs)
    else:
        # quick pairs are always behind the descriptor-matched pairs of their wave; the skipped ones are the thin view's
        for w in range(st["waves"]):
            q = [x[2] for x in trace if x[0] == w]
            assert q == sorted(q)
        thin = len(views) - 1
        assert all(x[1][1] == thin and x[2] for x in trace if x[4]) and st["too_few_matches"] == sum(x[4] for x in trace) >= 4
        assert st["guided_matching_runs"] == sum(1 for x in trace if x[2] and not x[4] and x[1] in edges)
        assert st["track_number"] > 1000
    if kw.get("rotation_guided"):
        assert st["quirk_only_guesses"] == 0 and st["poses_from_guess"] >= 0.9 * st["paths_found"]
