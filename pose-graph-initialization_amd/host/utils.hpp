// utils.hpp -- host utilities around the path (no GPU, no OpenCV): mirrors include/utils.h
//   RunningStatistics    utils.h:15-73  (mutex-guarded name -> (sum, count); the observability keys of
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

class RunningStatistics {
   public:
    void addValue(const std::string& name, double value) {
        std::lock_guard<std::mutex> l(mu);
        auto& e = values[name];
        e.first += value;
        e.second += 1;
    }
    double getSum(const std::string& name) const {
        std::lock_guard<std::mutex> l(mu);
        auto it = values.find(name);
        return it == values.end() ? 0.0 : it->second.first;
    }
    size_t getCount(const std::string& name) const {
        std::lock_guard<std::mutex> l(mu);
        auto it = values.find(name);
        return it == values.end() ? 0 : it->second.second;
    }
    double getAverage(const std::string& name) const {
        const size_t c = getCount(name);
        return c ? getSum(name) / (double)c : 0.0;
    }
    void print() const {  // utils.h:59-72
        std::lock_guard<std::mutex> l(mu);
        for (auto& kv : values)
            std::printf("%s\n\tAverage = %f\n\tTotal = %f\n\tCount = %zu\n", kv.first.c_str(),
                        kv.second.second ? kv.second.first / (double)kv.second.second : 0.0, kv.second.first,
                        kv.second.second);
    }

   protected:
    mutable std::mutex mu;
    std::map<std::string, std::pair<double, size_t>> values;
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
