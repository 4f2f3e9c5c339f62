// Minimal API-compatible subset of <SoapySDR/Time.hpp> (see Constants.h).
#pragma once
#include "Constants.h"

extern "C" {
long long SoapySDR_ticksToTimeNs(const long long ticks, const double rate);
long long SoapySDR_timeNsToTicks(const long long timeNs, const double rate);
}

namespace SoapySDR {
static inline long long ticksToTimeNs(const long long ticks, const double rate)
{
    return SoapySDR_ticksToTimeNs(ticks, rate);
}
static inline long long timeNsToTicks(const long long timeNs, const double rate)
{
    return SoapySDR_timeNsToTicks(timeNs, rate);
}
}  // namespace SoapySDR
