"""tracklets_oracle.py -- CPU ORACLE for the tracklet bookkeeping (test infrastructure, NOT product code).

Restates reconstruction::Tracklets::add (point_track.h:633-711) and ::getCorrespondences (point_track.h:568-631)
in plain Python, quirks included: id 0 doubles as "unseen" (:651-657), the result may hold max + 1 entries
(:626-627), a destination index defaults to 0 when the track lists the source view twice first (:608-622).
PARITY UNPINNED against the reference binary (it cannot be built here); the restatement follows the source line by line.
"""


class Tracklets:
    def __init__(self):
        self.n_ids = 0
        self.ids = {}            # (view, point) -> id
        self.tracks = []         # list of lists of (view, point)
        self.view_tracks = {}    # view -> [track idx] in insertion order (repeats possible)
        self.id_tracks = {}      # id -> [track idx]

    def _id(self, pair):
        v = self.ids.get(pair, 0)
        if v == 0:               # unseen OR the very first point ever registered
            v = self.n_ids
            self.n_ids += 1
            self.ids[pair] = v
        return v

    def add(self, src, dst, matches, mask):
        for (p1, p2, *_), keep in zip(matches, mask):
            if not keep:
                continue
            ps, pd = (src, int(p1)), (dst, int(p2))
            ts = self.id_tracks.setdefault(self._id(ps), [])
            td = self.id_tracks.setdefault(self._id(pd), [])
            n_dst = len(td)
            added = False
            for t in list(ts):
                if pd in self.tracks[t]:
                    continue
                self.view_tracks.setdefault(dst, []).append(t)
                self.tracks[t].append(pd)
                td.append(t)
                added = True
            for t in td[:n_dst]:
                if ps in self.tracks[t]:
                    continue
                self.view_tracks.setdefault(src, []).append(t)
                self.tracks[t].append(ps)
                ts.append(t)
                added = True
            if not added:
                idx = len(self.tracks)
                self.tracks.append([ps, pd])
                self.view_tracks.setdefault(src, []).append(idx)
                self.view_tracks.setdefault(dst, []).append(idx)
                ts.append(idx)
                td.append(idx)

    def get_correspondences(self, src, dst, max_n):
        out = []
        if src not in self.view_tracks or dst not in self.view_tracks:
            return out
        of_src = set(self.view_tracks[src])
        for t in self.view_tracks[dst]:
            if t not in of_src:
                continue
            a = b = found = 0
            for v, p in self.tracks[t]:
                if v == src:
                    a, found = p, found + 1
                elif v == dst:
                    b, found = p, found + 1
                if found == 2:
                    break
            out.append((a, b))
            if len(out) > max_n:
                break
        return out
