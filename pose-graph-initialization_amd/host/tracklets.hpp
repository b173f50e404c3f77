// tracklets.hpp -- multi-view tracklets feeding "quick matching" (SURVEY §8f-2).
//
// Observable behaviour of reconstruction::Tracklets (reference: point_track.h:541-711), which the scheduler relies on
// for functional parity: add() grows tracks from the inlier matches of an estimated edge; getCorrespondences() returns,
// for a later pair, the keypoints both views share a track with -- correspondences without descriptor matching.
// The behaviour is pinned by oracle/tracklets_oracle.py (tests/test_tracklets.py), including what callers can observe
// of the reference's quirks: a point id of 0 also means "never seen" (so the first point ever registered is registered
// again on its second visit and leaves its first tracks behind), getCorrespondences stops only after the result has
// EXCEEDED the maximum (max + 1 entries), and a track that lists the source view twice before the destination yields
// destination index 0.
//
// Design (not the reference's containers): the structure is an incidence store between points and tracks whose lists
// are append-only and only ever walked front to back.  Everything lives in flat arrays of fixed-size records:
//     PointRecord  the tracks holding a point: the first 3 inline, the rest in a chain of overflow cells
//     TrackRecord  the members (view, keypoint) of a track: the first 7 inline (one 64-byte cache line), rest chained
//     per view     one contiguous array of track indices (append order), and a direct keypoint -> point-id table
// so a match costs a handful of cache lines and no allocation (the common point sits in one or two tracks of a few
// members).  The work itself is a chain of dependent appends per match -- a match sees the tracks its predecessors
// just extended -- which is why this piece stays on the host: on the GPU it would be one lane chasing pointers
// through HBM (pipeline table).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <shared_mutex>
#include <tuple>
#include <utility>
#include <vector>

namespace reconstruction {

class Tracklets {
   public:
    typedef std::pair<size_t, size_t> Pair;  // (view index, keypoint index)
    typedef std::tuple<size_t, size_t, double> Match;

    explicit Tracklets(size_t viewNumber_ = 0) { views.reserve(viewNumber_); }

    // Correspondences of (source, destination) from shared tracks, in the order of the destination view's track list.
    void getCorrespondences(std::vector<Match>& matches_, const size_t& viewIdSource_, const size_t& viewIdDestination_,
                            const size_t& maximumCorrespondenceNumber_) const {
        std::shared_lock<std::shared_mutex> lock(readerWriterLock);
        if (!knowsView(viewIdSource_) || !knowsView(viewIdDestination_)) return;
        // which tracks touch the source view: one generation stamp per track (per calling thread)
        static thread_local std::vector<uint32_t> seen;
        static thread_local uint32_t generation = 0;
        if (seen.size() < tracks.size()) seen.resize(tracks.size(), 0u);
        if (++generation == 0u) {
            seen.assign(seen.size(), 0u);
            generation = 1u;
        }
        for (const uint32_t track : views[viewIdSource_].tracks) seen[track] = generation;
        const uint32_t vs = (uint32_t)viewIdSource_, vd = (uint32_t)viewIdDestination_;
        for (const uint32_t track : views[viewIdDestination_].tracks) {
            if (seen[track] != generation) continue;
            size_t pointSource = 0, pointDestination = 0;
            int hits = 0;
            forEachMember(track, [&](uint32_t view, uint32_t point) {
                if (view == vs) {
                    pointSource = point;
                    ++hits;
                } else if (view == vd) {
                    pointDestination = point;
                    ++hits;
                }
                return hits < 2;
            });
            matches_.emplace_back(pointSource, pointDestination, 0.0);
            if (matches_.size() > maximumCorrespondenceNumber_) break;
        }
    }

    // Registers the masked-in matches of the edge (source, destination).
    void add(const size_t& imageIdxSource_, const size_t& imageIdxDestination_, const std::vector<Match>& matches_,
             const std::vector<unsigned char>& inlierMask_) {
        std::unique_lock<std::shared_mutex> lock(readerWriterLock);
        const uint32_t vs = (uint32_t)imageIdxSource_, vd = (uint32_t)imageIdxDestination_;
        std::vector<uint32_t> snapshot;  // the destination point's tracks as they were before the match
        for (size_t k = 0; k < matches_.size(); ++k) {
            if (!inlierMask_[k]) continue;
            const uint32_t ps = (uint32_t)std::get<0>(matches_[k]), pd = (uint32_t)std::get<1>(matches_[k]);
            const uint32_t idS = pointId(vs, ps), idD = pointId(vd, pd);
            snapshot.clear();
            forEachTrackOf(idD, [&](uint32_t track) { snapshot.push_back(track); });
            bool extended = false;
            // every track of the source point learns the destination point ...
            forEachTrackOf(idS, [&](uint32_t track) {
                if (hasMember(track, vd, pd)) return;
                touch(vd).tracks.push_back(track);
                pushMember(track, vd, pd);
                pushTrackOf(idD, track);
                extended = true;
            });
            // ... and every earlier track of the destination point learns the source point
            for (const uint32_t track : snapshot) {
                if (hasMember(track, vs, ps)) continue;
                touch(vs).tracks.push_back(track);
                pushMember(track, vs, ps);
                pushTrackOf(idS, track);
                extended = true;
            }
            if (extended) continue;
            // neither side had a track to extend: the match starts one
            const uint32_t track = (uint32_t)tracks.size();
            tracks.emplace_back();
            pushMember(track, vs, ps);
            pushMember(track, vd, pd);
            touch(vs).tracks.push_back(track);
            touch(vd).tracks.push_back(track);
            pushTrackOf(idS, track);
            pushTrackOf(idD, track);
        }
    }

    size_t trackNumber() const { return tracks.size(); }
    // members of one track in insertion order (diagnostics / tests)
    std::vector<Pair> track(size_t index) const {
        std::vector<Pair> out;
        forEachMember((uint32_t)index, [&](uint32_t view, uint32_t point) {
            out.emplace_back(view, point);
            return true;
        });
        return out;
    }

   private:
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    static constexpr uint32_t kInlineTracks = 3, kInlineMembers = 7;
    struct PointRecord {  // 24 bytes
        uint32_t count = 0;
        uint32_t first[kInlineTracks] = {0, 0, 0};
        uint32_t head = kNone, tail = kNone;  // overflow chain in `spill`
    };
    struct TrackRecord {  // 64 bytes: one cache line
        uint32_t count = 0;
        uint32_t head = kNone;  // overflow chain in `memberSpill`
        struct {
            uint32_t view, point;
        } first[kInlineMembers] = {};
    };
    struct ViewRecord {
        std::vector<uint32_t> tracks;  // tracks touching the view, once per member of that view, in append order
        std::vector<uint32_t> ids;     // keypoint -> point id (0 = never seen; see pointId)
        bool known = false;            // becomes true with the first entry of `tracks`
    };
    struct SpillCell {
        uint32_t next, value;
    };
    struct MemberCell {
        uint32_t next, view, point;
    };

    bool knowsView(size_t v) const { return v < views.size() && views[v].known; }
    ViewRecord& touch(uint32_t v) {
        if (v >= views.size()) views.resize((size_t)v + 1);
        views[v].known = true;
        return views[v];
    }
    template <class F>
    void forEachTrackOf(uint32_t id, F visit) const {  // `visit` may append to OTHER points' lists
        const uint32_t n = points[id].count;
        for (uint32_t q = 0; q < n && q < kInlineTracks; ++q) visit(points[id].first[q]);
        uint32_t c = points[id].head;
        for (uint32_t q = kInlineTracks; q < n; ++q, c = spill[c].next) visit(spill[c].value);
    }
    void pushTrackOf(uint32_t id, uint32_t track) {
        PointRecord& r = points[id];
        if (r.count < kInlineTracks) {
            r.first[r.count] = track;
        } else {
            const uint32_t c = (uint32_t)spill.size();
            spill.push_back(SpillCell{kNone, track});
            if (r.tail == kNone) r.head = c; else spill[r.tail].next = c;
            r.tail = c;
        }
        ++r.count;
    }
    template <class F>
    void forEachMember(uint32_t track, F visit) const {  // visit returns false to stop
        const TrackRecord& t = tracks[track];
        for (uint32_t q = 0; q < t.count && q < kInlineMembers; ++q)
            if (!visit(t.first[q].view, t.first[q].point)) return;
        uint32_t c = t.head;
        for (uint32_t q = kInlineMembers; q < t.count; ++q, c = memberSpill[c].next)
            if (!visit(memberSpill[c].view, memberSpill[c].point)) return;
    }
    bool hasMember(uint32_t track, uint32_t view, uint32_t point) const {
        bool found = false;
        forEachMember(track, [&](uint32_t v, uint32_t p) {
            found = (v == view && p == point);
            return !found;
        });
        return found;
    }
    void pushMember(uint32_t track, uint32_t view, uint32_t point) {
        TrackRecord& t = tracks[track];
        if (t.count < kInlineMembers) {
            t.first[t.count].view = view;
            t.first[t.count].point = point;
        } else {  // chain order = insertion order: walk to the end (overflow chains are short and rare)
            const uint32_t c = (uint32_t)memberSpill.size();
            memberSpill.push_back(MemberCell{kNone, view, point});
            if (t.head == kNone) {
                t.head = c;
            } else {
                uint32_t e = t.head;
                while (memberSpill[e].next != kNone) e = memberSpill[e].next;
                memberSpill[e].next = c;
            }
        }
        ++t.count;
    }
    // id of a point; ids count up from 0 and an id of 0 is indistinguishable from "never seen", so whoever holds 0
    // receives a fresh id on the next visit
    uint32_t pointId(uint32_t view, uint32_t point) {
        if (view >= views.size()) views.resize((size_t)view + 1);
        std::vector<uint32_t>& ids = views[view].ids;
        if (point >= ids.size()) ids.resize(std::max<size_t>((size_t)point + 1, ids.size() * 2), 0u);
        if (ids[point] == 0) {
            ids[point] = nextId++;
            if (points.size() < nextId) points.resize(std::max<size_t>(nextId, points.size() * 2));
        }
        return ids[point];
    }

    mutable std::shared_mutex readerWriterLock;
    uint32_t nextId = 0;
    std::vector<PointRecord> points;       // by point id
    std::vector<TrackRecord> tracks;       // by track index
    std::vector<ViewRecord> views;         // by view index
    std::vector<SpillCell> spill;          // overflow of PointRecord::first
    std::vector<MemberCell> memberSpill;   // overflow of TrackRecord::first
};

}  // namespace reconstruction
