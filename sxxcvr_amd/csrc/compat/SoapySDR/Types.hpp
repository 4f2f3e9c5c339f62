// Minimal API-compatible subset of <SoapySDR/Types.hpp> (see Constants.h).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "Constants.h"

namespace SoapySDR {

typedef std::map<std::string, std::string> Kwargs;
typedef std::vector<Kwargs> KwargsList;

Kwargs KwargsFromString(const std::string &markup);
std::string KwargsToString(const Kwargs &args);

class Range {
public:
    Range(void) : _min(0), _max(0), _step(0) {}
    Range(const double minimum, const double maximum, const double step = 0.0)
        : _min(minimum), _max(maximum), _step(step) {}
    double minimum(void) const { return _min; }
    double maximum(void) const { return _max; }
    double step(void) const { return _step; }

private:
    double _min, _max, _step;
};

typedef std::vector<Range> RangeList;

}  // namespace SoapySDR
