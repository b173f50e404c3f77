// utils.hpp -- host utilities around the path (no GPU, no OpenCV): mirrors include/utils.h
//   RunningStatistics    utils.h:15-73  (mutex-guarded time and count tables; the observability keys of
//                        processImages: "[A*]", "[Pose estimation]", ... SURVEY.md §5)
//   load1DSfMImageList   utils.h:122-182 ("images/<name> 0 <focal>" lines of list_with_focals.txt)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

namespace reconstruction {

// Two tables like the reference's: elapsed seconds per stage and event counts, each entry = (sum, number of
// additions).  The key strings used by run / processFeatures are the reference's own (pose_graph_builder.h:505-518,
// 545-546, 599-602, 636-638, 684-686, 698-699): "[Quick matching]", "[Matching]", "[A*]", "[Pose estimation]",
// "[Epipolar Hashing]", "[Visibility update]" and their " Runs" / " Touched nodes" / " Paths tested" /
// " Inlier number" / " Correspondences added" counters.
class RunningStatistics {
   public:
    typedef std::map<std::string, std::pair<double, size_t>> TimeTable;
    typedef std::map<std::string, std::pair<size_t, size_t>> CountTable;
    void addTime(const std::string& kPropertyName_, double value_, size_t events_ = 1) {
        std::lock_guard<std::mutex> l(guard);
        auto& e = timeValues[kPropertyName_];
        e.first += value_;
        e.second += events_;
    }
    void addCount(const std::string& kPropertyName_, size_t value_, size_t events_ = 1) {
        std::lock_guard<std::mutex> l(guard);
        auto& e = countValues[kPropertyName_];
        e.first += value_;
        e.second += events_;
    }
    TimeTable getTimes() const {
        std::lock_guard<std::mutex> l(guard);
        return timeValues;
    }
    CountTable getCounts() const {
        std::lock_guard<std::mutex> l(guard);
        return countValues;
    }
    size_t getCount(const std::string& kPropertyName_) const {
        std::lock_guard<std::mutex> l(guard);
        auto it = countValues.find(kPropertyName_);
        return it == countValues.end() ? 0 : it->second.first;
    }
    double getTime(const std::string& kPropertyName_) const {
        std::lock_guard<std::mutex> l(guard);
        auto it = timeValues.find(kPropertyName_);
        return it == timeValues.end() ? 0.0 : it->second.first;
    }
    std::pair<double, bool> getAverageTime(const std::string& kPropertyName_) const {
        std::lock_guard<std::mutex> l(guard);
        auto it = timeValues.find(kPropertyName_);
        if (it == timeValues.end() || it->second.second == 0) return {0.0, false};
        return {it->second.first / (double)it->second.second, true};
    }
    void clear() {
        std::lock_guard<std::mutex> l(guard);
        timeValues.clear();
        countValues.clear();
    }
    void print() const {  // layout of utils.h:59-72
        const TimeTable t = getTimes();
        const CountTable c = getCounts();
        std::printf("Statistics:\n");
        for (auto& kv : t)
            std::printf("\tAverage '%s' time = %f seconds\n", kv.first.c_str(), kv.second.second ? kv.second.first / (double)kv.second.second : 0.0);
        std::printf("\t------------------\n");
        for (auto& kv : t) std::printf("\tTotal '%s' time = %f seconds\n", kv.first.c_str(), kv.second.first);
        std::printf("\t------------------\n");
        for (auto& kv : c) std::printf("\tNumber of '%s' = %d\n", kv.first.c_str(), (int)kv.second.first);
    }

   protected:
    mutable std::mutex guard;
    TimeTable timeValues;
    CountTable countValues;
};

// utils.h:122-182.  results_: (image name without the "images/" prefix, focal length, width, height).
// Lines without a third token keep focal length 0 (the reference leaves it uninitialised there);
// image sizes are not probed here (the reference reads them with cv::imread) and stay 0.
inline bool load1DSfMImageList(const std::string& kListPath_, size_t& totalImageNumber_,
                               std::vector<std::tuple<std::string, double, double, double>>& results_) {
    totalImageNumber_ = 0;
    std::ifstream file(kListPath_);
    if (!file.is_open()) return false;
    std::string line;
    while (std::getline(file, line)) {
        ++totalImageNumber_;
        size_t counter = 0;
        std::istringstream iss(line);
        std::string imageName, s;
        double focalLength = 0.0;
        while (iss >> s) {
            switch (counter++) {
                case 0:
                    imageName = s.size() >= 7 ? s.substr(7, s.size() - 7) : std::string();
                    break;
                case 1:
                    break;
                case 2:
                    focalLength = std::atof(s.c_str());
                    break;
            }
        }
        results_.emplace_back(imageName, focalLength, 0.0, 0.0);
    }
    return true;
}

}  // namespace reconstruction
