// utils.hpp -- host utilities around the path (no GPU, no OpenCV): mirrors include/utils.h
//   RunningStatistics    utils.h:15-73  (mutex-guarded time and count tables; the observability keys of
//                        processImages: "[A*]", "[Pose estimation]", ... SURVEY.md §5)
//   load1DSfMImageList   utils.h:122-182 ("images/<name> 0 <focal>" lines of list_with_focals.txt)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
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

// The 1DSfM image list, "list_with_focals.txt" (format: utils.h:122-182): one image per line,
//     images/<name> [<flag> <focal length>]
// results_ gets (name without the "images/" directory, focal length, width, height) per image; totalImageNumber_ the
// number of images listed.  Parsed strictly, which the reference does not do: a line without
// a focal length yields focal 0 -- the reference leaves the value uninitialised there; a focal length that is not a
// number makes the whole load fail instead of slipping through atof; fields after the third are ignored and an empty
// line is a record of its own (empty name, focal 0), both as in the reference, so that line index == view id holds.  A name without the "images/" prefix is kept whole (the reference cuts seven characters blindly).
// Image sizes are not probed here (the reference reads them with cv::imread) and stay 0.
inline bool load1DSfMImageList(const std::string& kListPath_, size_t& totalImageNumber_,
                               std::vector<std::tuple<std::string, double, double, double>>& results_) {
    totalImageNumber_ = 0;
    std::ifstream file(kListPath_, std::ios::binary);
    if (!file.is_open()) return false;
    const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    static const char kDir[] = "images/";
    const auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r'; };
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string field[3];  // name, flag, focal length
        size_t nfields = 0, p = pos;
        while (p < eol) {
            while (p < eol && blank(text[p])) ++p;
            if (p == eol) break;
            size_t q = p;
            while (q < eol && !blank(text[q])) ++q;
            if (nfields < 3) field[nfields++] = text.substr(p, q - p);  // further fields are ignored, as in the reference (:147-160)
            p = q;
        }
        pos = eol + 1;
        // ONE RECORD PER LINE, blank lines included (utils.h:136-168 counts and pushes every line it reads): the view id of
        // an image is its line index, and the similarity matrix's rows are numbered the same way -- dropping a blank line
        // would shift every later view against the matrix.  A blank line yields an empty name with focal length 0.
        double focal = 0.0;
        if (nfields == 3) {
            char* end = nullptr;
            focal = std::strtod(field[2].c_str(), &end);
            if (end == field[2].c_str() || *end != '\0' || !(focal >= 0.0)) return false;
        }
        const bool prefixed = field[0].compare(0, sizeof kDir - 1, kDir) == 0;
        results_.emplace_back(prefixed ? field[0].substr(sizeof kDir - 1) : field[0], focal, 0.0, 0.0);
        ++totalImageNumber_;
    }
    return true;
}

}  // namespace reconstruction
