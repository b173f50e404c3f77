"""Device-resident tracklets (pgi_tracklets_*, csrc/pgi_tracklets.hip) against the Python restatement of
point_track.h:541-711 (oracle/tracklets_oracle.py): track indices, member order, every view's track list and the
correspondences must be those of the reference applied one match at a time."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tracklets_oracle as TO  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder.engine import Engine
    e = Engine()
    yield e
    e.close()


def _check_state(dev, ref, n_views, max_ns=(0, 3, 1000)):
    info = dev.info()
    assert info["tracks"] == len(ref.tracks)
    assert info["events"] == sum(len(t) for t in ref.tracks)
    queries = [(a, b) for a in range(n_views) for b in range(n_views) if a != b]
    for max_n in max_ns:
        got = dev.get_correspondences_batch(queries, max_n)
        for (a, b), g in zip(queries, got):
            exp = ref.get_correspondences(a, b, max_n)
            assert [tuple(r) for r in g.tolist()] == [tuple(r) for r in exp], (a, b, max_n)


def _check_tracks(dev, ref, sample):
    for i in sample:
        assert dev.track(i) == [tuple(m) for m in ref.tracks[i]], i


def test_walkthrough_and_quirks(eng):
    """tests/test_tracklets.py's walkthrough on the device: extension across views, a masked-out match, the max + 1
    cut-off and the id-0 quirk (the first point ever registered starts over on its second visit)."""
    from pyposegraphbuilder.engine import DeviceTracklets
    dev, ref = DeviceTracklets(eng, 8), TO.Tracklets()
    m01 = [(10, 20), (11, 21), (12, 22), (13, 23)]
    m12 = [(20, 30), (21, 31), (22, 32), (99, 98)]
    for s, d, m, k in [(0, 1, m01, [1, 1, 1, 0]), (1, 2, m12, [1, 1, 1, 1])]:
        dev.add(s, d, m, k)
        ref.add(s, d, m, k)
    assert dev.get_correspondences(0, 2, 100).tolist() == [[10, 30], [11, 31], [12, 32]]
    assert len(dev.get_correspondences(0, 2, 1)) == 2
    assert len(dev.get_correspondences(0, 7, 100)) == 0
    dev.add(0, 1, m01[:1], [1])
    ref.add(0, 1, m01[:1], [1])
    assert dev.info()["tracks"] == 5 == len(ref.tracks)
    _check_state(dev, ref, 8)
    _check_tracks(dev, ref, range(5))
    dev.close()


@pytest.mark.parametrize("batched", [False, True])
def test_random_sequences_match_oracle(eng, batched):
    """Random adds with many-to-one matches, wrong matches and masks; one add per call, or several calls per batch
    (the scheduler's wave commit)."""
    from pyposegraphbuilder.engine import DeviceTracklets
    rng = np.random.default_rng(5 + batched)
    for trial in range(4):
        views, pts = 9, 50
        truth = rng.integers(0, 30, (views, pts))
        dev, ref = DeviceTracklets(eng, views), TO.Tracklets()
        pending = []
        for step in range(70):
            s, d = (int(v) for v in rng.choice(views, 2, replace=False))
            m = int(rng.integers(0, 30))
            matches = []
            for p in rng.integers(0, pts, m):
                cand = np.nonzero(truth[d] == truth[s, p])[0]
                q = int(rng.choice(cand)) if len(cand) and rng.random() < 0.85 else int(rng.integers(0, pts))
                matches.append((int(p), q))
            mask = [int(v) for v in rng.random(m) < 0.8]
            ref.add(s, d, matches, mask)
            pending.append((s, d, matches, mask if trial % 2 == 0 else (None if all(mask) else mask)))
            if not batched or len(pending) == 6:
                dev.add_batch(pending)
                pending = []
                if step % 10 == 9 or batched:
                    _check_state(dev, ref, views, max_ns=(2, 1000))
        dev.add_batch(pending)
        _check_state(dev, ref, views)
        _check_tracks(dev, ref, range(0, len(ref.tracks), 7))
        dev.close()


def test_dense_repeats_and_first_point_returning(eng):
    """Destination keypoints repeat inside one add() (guided matching keeps the best destination per source), the
    first point ever registered comes back inside the same batch and in later ones."""
    from pyposegraphbuilder.engine import DeviceTracklets
    rng = np.random.default_rng(23)
    for trial in range(3):
        views, pts = 12, 60
        dev, ref = DeviceTracklets(eng, views), TO.Tracklets()
        first = None
        for wave in range(12):
            calls = []
            for _ in range(10):
                s, d = (int(v) for v in rng.choice(views, 2, replace=False))
                m = int(rng.integers(1, 40))
                matches = [(int(a), int(b)) for a, b in zip(rng.permutation(pts)[:m], rng.integers(0, pts // (1 + trial), m))]
                if first is None:
                    first = (s, d, matches[0])
                elif (s, d) == first[:2] or rng.random() < 0.2:
                    if (s, d) == first[:2]:
                        matches.insert(int(rng.integers(0, len(matches))), first[2])
                mask = [int(v) for v in rng.random(len(matches)) < 0.9]
                calls.append((s, d, matches, mask))
                ref.add(s, d, matches, mask)
            dev.add_batch(calls)
            _check_state(dev, ref, views, max_ns=(5, 1000))
        _check_tracks(dev, ref, range(0, len(ref.tracks), 11))
        dev.close()


@pytest.fixture(scope="module")
def scheduler_scene():
    """24 views of 1500 physical points; three waves of 40 pairs x ~800 inlier matches, and the oracle's state after each."""
    import copy
    rng = np.random.default_rng(99)
    views, pts = 24, 1500
    perm = [rng.permutation(pts) for _ in range(views)]       # keypoint index of physical point j in view v
    ref = TO.Tracklets()
    pairs = [(a, b) for a in range(views) for b in range(a + 1, views)]
    order = rng.permutation(len(pairs))
    waves = []
    for w in range(3):
        calls = []
        for pi in order[w * 40:(w + 1) * 40]:
            a, b = pairs[pi]
            vis = np.nonzero(rng.random(pts) < 0.6)[0]
            src = perm[a][vis]
            dst = perm[b][vis].copy()
            wrong = rng.random(len(vis)) < 0.05
            dst[wrong] = rng.integers(0, pts, int(wrong.sum()))
            m = np.stack([src, dst], 1)
            mask = (rng.random(len(vis)) < 0.9).astype(np.uint8)
            calls.append((a, b, m, mask))
            ref.add(a, b, [tuple(r) for r in m.tolist()], mask.tolist())
        waves.append((calls, copy.deepcopy(ref) if w == 1 else None))
    return views, waves, ref


@pytest.mark.parametrize("event_cap", [None, 3000])
def test_scheduler_sized_batch(eng, scheduler_scene, event_cap, monkeypatch):
    """Waves the size the pipeline commits, each one batch on top of the committed store; compared with the oracle on
    every view pair.  With PGI_TRACKLETS_EVENT_CAP=3000 the event buffer starts far too small and grows several times in
    the middle of every batch."""
    from pyposegraphbuilder.engine import DeviceTracklets
    if event_cap:
        monkeypatch.setenv("PGI_TRACKLETS_EVENT_CAP", str(event_cap))
    views, waves, ref = scheduler_scene
    dev = DeviceTracklets(eng, views)
    for calls, snapshot in waves:
        dev.add_batch(calls)
        assert dev.info()["rounds"] < 1000
        if snapshot is not None:
            _check_state(dev, snapshot, views, max_ns=(30000,))
    _check_state(dev, ref, views, max_ns=(100, 30000))
    _check_tracks(dev, ref, range(0, len(ref.tracks), 997))
    dev.close()


@pytest.mark.parametrize("list_cap", [None, 48])
def test_long_track_lists(eng, list_cap, monkeypatch):
    """Few keypoints, many views, 30 % wrong matches: every keypoint ends up in more than a hundred tracks, so lists span
    several 64-entry chunks and more than 64 runs per batch.  With PGI_TRACKLETS_LIST_CAP=48 the usual configuration
    cannot stage them and the batch finishes in the large-list configuration."""
    from pyposegraphbuilder.engine import DeviceTracklets
    if list_cap:
        monkeypatch.setenv("PGI_TRACKLETS_LIST_CAP", str(list_cap))
    rng = np.random.default_rng(1)
    V, K = 30, 12
    dev, ref = DeviceTracklets(eng, V), TO.Tracklets()
    pairs = [(a, b) for a in range(V) for b in range(a + 1, V)]
    order = rng.permutation(len(pairs))
    calls = []
    for n, pi in enumerate(order):
        a, b = pairs[pi]
        m = [(p, p if rng.random() > 0.3 else int(rng.integers(0, K))) for p in range(K) if rng.random() < 0.7]
        ref.add(a, b, m, [1] * len(m))
        calls.append((a, b, m, None))
        if len(calls) == 145:
            dev.add_batch(calls)
            calls = []
            _check_state(dev, ref, V, max_ns=(1000000,))
    assert not calls
    assert max(len(v) for v in ref.id_tracks.values()) > 100
    _check_tracks(dev, ref, range(len(ref.tracks)))
    dev.close()


def test_store_is_released_with_its_engine():
    """Closing the engine first must not leave the store pointing at a dead context."""
    from pyposegraphbuilder.engine import DeviceTracklets, Engine
    e = Engine()
    dev = DeviceTracklets(e, 3)
    dev.add(0, 1, [(1, 2), (3, 4)], None)
    e.close()          # destroys the store, then the context
    assert dev._t is None
    dev.close()        # no-op
    del dev, e


def test_argument_errors(eng):
    from pyposegraphbuilder import _lib as L
    from pyposegraphbuilder.engine import DeviceTracklets
    dev = DeviceTracklets(eng, 4)
    with pytest.raises(L.PgiError):
        dev.add(0, 4, [(1, 2)], None)           # view out of range
    with pytest.raises(L.PgiError):
        dev.add(2, 2, [(1, 2)], None)           # source == destination
    dev.add(0, 1, [], None)                     # empty add: nothing happens
    assert dev.info() == {"tracks": 0, "events": 0, "rounds": 0}
    assert len(dev.get_correspondences(0, 1, 10)) == 0
    dev.close()
