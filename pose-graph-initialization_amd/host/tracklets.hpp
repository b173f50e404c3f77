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
        std::unordered_set<size_t> ofSource(it->second.begin(), it->second.end());
        for (const size_t trackIdx : jt->second) {
            if (!ofSource.count(trackIdx)) continue;
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
        for (size_t k = 0; k < matches_.size(); ++k) {
            if (!inlierMask_[k]) continue;
            const Pair pairSource(imageIdxSource_, std::get<0>(matches_[k]));
            const Pair pairDestination(imageIdxDestination_, std::get<1>(matches_[k]));
            size_t& idS = pointPairs[pairSource];
            if (idS == 0) idS = pointPairNumber++;
            const size_t idSource = idS;  // (the reference holds references; std::map never invalidates them either)
            size_t& idD = pointPairs[pairDestination];
            if (idD == 0) idD = pointPairNumber++;
            const size_t idDestination = idD;
            std::vector<size_t>& tracksSource = tmpPairToTracks[idSource];
            std::vector<size_t>& tracksDestination = tmpPairToTracks[idDestination];
            const size_t trackNumDestination = tracksDestination.size();
            bool added = false;
            for (size_t q = 0; q < tracksSource.size(); ++q) {  // tracksSource does not grow in this loop
                const size_t trackIdx = tracksSource[q];
                std::vector<Pair>& track = tmpTracks[trackIdx];
                if (std::find(track.begin(), track.end(), pairDestination) != track.end()) continue;
                tmpViewToTracks[imageIdxDestination_].emplace_back(trackIdx);
                track.emplace_back(pairDestination);
                tracksDestination.emplace_back(trackIdx);
                added = true;
            }
            for (size_t q = 0; q < trackNumDestination; ++q) {
                const size_t trackIdx = tracksDestination[q];
                std::vector<Pair>& track = tmpTracks[trackIdx];
                if (std::find(track.begin(), track.end(), pairSource) != track.end()) continue;
                tmpViewToTracks[imageIdxSource_].emplace_back(trackIdx);
                track.emplace_back(pairSource);
                tracksSource.emplace_back(trackIdx);
                added = true;
            }
            if (!added) {
                const size_t idx = tmpTracks.size();
                tmpTracks.emplace_back(std::vector<Pair>{pairSource, pairDestination});
                tmpViewToTracks[imageIdxSource_].emplace_back(idx);
                tmpViewToTracks[imageIdxDestination_].emplace_back(idx);
                tracksSource.emplace_back(idx);
                tracksDestination.emplace_back(idx);
            }
        }
    }

    size_t trackNumber() const { return tmpTracks.size(); }
    const std::vector<std::vector<Pair>>& tracks() const { return tmpTracks; }

private:
    mutable std::shared_mutex readerWriterLock;
    size_t pointPairNumber;
    std::map<Pair, size_t> pointPairs;                           // (view, point) -> id
    std::vector<std::vector<Pair>> tmpTracks;                    // track -> its (view, point) members
    std::unordered_map<size_t, std::vector<size_t>> tmpViewToTracks;  // view -> tracks touching it (with repeats)
    std::map<size_t, std::vector<size_t>> tmpPairToTracks;       // id -> tracks containing that point
};

}  // namespace reconstruction
