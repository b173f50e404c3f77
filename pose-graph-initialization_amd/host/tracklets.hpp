// tracklets.hpp -- multi-view tracklets feeding "quick matching" (SURVEY §8f-2).
//
// Mirrors reconstruction::Tracklets (point_track.h:541-711 of the reference): add() grows tracks from the inlier
// matches of an estimated edge, getCorrespondences() returns, for a later pair, the points both views share a
// track with -- correspondences without descriptor matching.  Written from the behaviour, including two quirks that
// callers can observe (kept on purpose, tests pin them against oracle/tracklets_oracle.py):
//   * point ids start at 0 and 0 also means "not seen yet" (point_track.h:651-657), so the very first point ever
//     added is re-registered under a fresh id whenever it shows up again and loses its earlier tracks;
//   * getCorrespondences stops after the size EXCEEDS the maximum (:626-627), i.e. returns up to max + 1 entries,
//     and fills the destination index with 0 when a track holds the source view twice before the destination (:608-622).
// Everything iterates over vectors in insertion order, so results are deterministic; the reader/writer lock of the
// reference is kept because PoseGraphBuilder::run may query from worker threads.
#pragma once
#include <algorithm>
#include <cstddef>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace reconstruction {

class Tracklets {
public:
    typedef std::pair<size_t, size_t> Pair;  // (view index, keypoint index)
    typedef std::tuple<size_t, size_t, double> Match;

    explicit Tracklets(size_t viewNumber_ = 0) : pointPairNumber(0) {
        tmpViewToTracks.reserve(viewNumber_);
    }

    // point_track.h:568-631
    void getCorrespondences(std::vector<Match>& matches_, const size_t& viewIdSource_, const size_t& viewIdDestination_,
                            const size_t& maximumCorrespondenceNumber_) const {
        std::shared_lock<std::shared_mutex> lock(readerWriterLock);
        const auto it = tmpViewToTracks.find(viewIdSource_);
        if (it == tmpViewToTracks.end()) return;
        const auto jt = tmpViewToTracks.find(viewIdDestination_);
        if (jt == tmpViewToTracks.end()) return;
        // membership of a track in the source view's list: a per-thread stamp array instead of the reference's
        // unordered_set (same answers; the lists reach 10^5 entries and this lookup dominated the host time)
        static thread_local std::vector<unsigned> stamp;
        static thread_local unsigned epoch = 0;
        if (stamp.size() < tmpTracks.size()) stamp.resize(tmpTracks.size(), 0u);
        if (++epoch == 0u) { std::fill(stamp.begin(), stamp.end(), 0u); epoch = 1u; }
        for (const size_t trackIdx : it->second) stamp[trackIdx] = epoch;
        for (const size_t trackIdx : jt->second) {
            if (stamp[trackIdx] != epoch) continue;
            Match m(0, 0, 0.0);
            int found = 0;
            for (const Pair& p : tmpTracks[trackIdx]) {
                if (p.first == viewIdSource_) { std::get<0>(m) = p.second; ++found; }
                else if (p.first == viewIdDestination_) { std::get<1>(m) = p.second; ++found; }
                if (found == 2) break;
            }
            matches_.emplace_back(m);
            if (matches_.size() > maximumCorrespondenceNumber_) break;
        }
    }

    // point_track.h:633-711
    void add(const size_t& imageIdxSource_, const size_t& imageIdxDestination_, const std::vector<Match>& matches_,
             const std::vector<unsigned char>& inlierMask_) {
        std::unique_lock<std::shared_mutex> lock(readerWriterLock);
        // (unordered_map never invalidates references to its elements; a view's list is created on first use, as
        // operator[] does in the reference)
        std::vector<size_t>* viewTracksSource = nullptr;
        std::vector<size_t>* viewTracksDestination = nullptr;
        auto ofSource = [&]() -> std::vector<size_t>& {
            if (!viewTracksSource) viewTracksSource = &tmpViewToTracks[imageIdxSource_];
            return *viewTracksSource;
        };
        auto ofDestination = [&]() -> std::vector<size_t>& {
            if (!viewTracksDestination) viewTracksDestination = &tmpViewToTracks[imageIdxDestination_];
            return *viewTracksDestination;
        };
        for (size_t k = 0; k < matches_.size(); ++k) {
            if (!inlierMask_[k]) continue;
            const Pair pairSource(imageIdxSource_, std::get<0>(matches_[k]));
            const Pair pairDestination(imageIdxDestination_, std::get<1>(matches_[k]));
            const size_t idSource = idOf(pairSource), idDestination = idOf(pairDestination);
            if (tmpPairToTracks.size() < pointPairNumber) tmpPairToTracks.resize(pointPairNumber);  // before taking references
            std::vector<size_t>& tracksSource = tmpPairToTracks[idSource];
            std::vector<size_t>& tracksDestination = tmpPairToTracks[idDestination];
            const size_t trackNumDestination = tracksDestination.size();
            bool added = false;
            for (size_t q = 0; q < tracksSource.size(); ++q) {  // tracksSource does not grow in this loop
                const size_t trackIdx = tracksSource[q];
                std::vector<Pair>& track = tmpTracks[trackIdx];
                if (std::find(track.begin(), track.end(), pairDestination) != track.end()) continue;
                ofDestination().emplace_back(trackIdx);
                track.emplace_back(pairDestination);
                tracksDestination.emplace_back(trackIdx);
                added = true;
            }
            for (size_t q = 0; q < trackNumDestination; ++q) {
                const size_t trackIdx = tracksDestination[q];
                std::vector<Pair>& track = tmpTracks[trackIdx];
                if (std::find(track.begin(), track.end(), pairSource) != track.end()) continue;
                ofSource().emplace_back(trackIdx);
                track.emplace_back(pairSource);
                tracksSource.emplace_back(trackIdx);
                added = true;
            }
            if (!added) {
                const size_t idx = tmpTracks.size();
                tmpTracks.emplace_back(std::vector<Pair>{pairSource, pairDestination});
                ofSource().emplace_back(idx);
                ofDestination().emplace_back(idx);
                tracksSource.emplace_back(idx);
                tracksDestination.emplace_back(idx);
            }
        }
    }

    size_t trackNumber() const { return tmpTracks.size(); }
    const std::vector<std::vector<Pair>>& tracks() const { return tmpTracks; }

private:
    // point_track.h:651-657: id 0 doubles as "not seen yet", so the first point ever is re-registered on every visit
    size_t idOf(const Pair& p) {
        size_t& id = pointPairs[(static_cast<unsigned long long>(p.first) << 32) ^ static_cast<unsigned long long>(p.second)];
        if (id == 0) id = pointPairNumber++;
        return id;
    }
    mutable std::shared_mutex readerWriterLock;
    size_t pointPairNumber;
    std::unordered_map<unsigned long long, size_t> pointPairs;   // (view, point) -> id
    std::vector<std::vector<Pair>> tmpTracks;                    // track -> its (view, point) members
    std::unordered_map<size_t, std::vector<size_t>> tmpViewToTracks;  // view -> tracks touching it (with repeats)
    std::vector<std::vector<size_t>> tmpPairToTracks;            // id -> tracks containing that point (ids are dense)
};

}  // namespace reconstruction
