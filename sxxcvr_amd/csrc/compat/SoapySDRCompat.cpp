// Implementation of the compat SoapySDR subset (registry, logger, time, Device
// defaults).  Only built when the real SoapySDR library is absent.
#include <SoapySDR/Device.hpp>
#include <SoapySDR/Logger.hpp>
#include <SoapySDR/Registry.hpp>
#include <SoapySDR/Time.hpp>

#include <cmath>
#include <cstdio>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace {

struct Entry {
    SoapySDR::FindFunction find;
    SoapySDR::MakeFunction make;
};

std::map<std::string, Entry> &table()
{
    static std::map<std::string, Entry> t;
    return t;
}

std::mutex &log_mutex()
{
    static std::mutex m;
    return m;
}

void default_handler(const SoapySDRLogLevel level, const char *message)
{
    static const char *names[] = {"", "FATAL", "CRITICAL", "ERROR", "WARNING", "NOTICE", "INFO", "DEBUG", "TRACE", "SSI"};
    fprintf(stderr, "[%s] %s\n", names[level <= 9 ? level : 0], message);
}

SoapySDRLogHandler g_handler = default_handler;
SoapySDRLogLevel g_level = SOAPY_SDR_INFO;

}  // namespace

// ---- Types ------------------------------------------------------------------
SoapySDR::Kwargs SoapySDR::KwargsFromString(const std::string &markup)
{
    Kwargs out;
    std::string item;
    std::stringstream ss(markup);
    while (std::getline(ss, item, ',')) {
        const size_t eq = item.find('=');
        auto trim = [](std::string s) {
            const size_t b = s.find_first_not_of(" \t");
            const size_t e = s.find_last_not_of(" \t");
            return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
        };
        if (eq == std::string::npos) {
            if (!trim(item).empty()) out[trim(item)] = "";
        } else {
            out[trim(item.substr(0, eq))] = trim(item.substr(eq + 1));
        }
    }
    return out;
}

std::string SoapySDR::KwargsToString(const Kwargs &args)
{
    std::string s;
    for (const auto &kv : args) {
        if (!s.empty()) s += ", ";
        s += kv.first + "=" + kv.second;
    }
    return s;
}

// ---- Registry ---------------------------------------------------------------
SoapySDR::Registry::Registry(const std::string &name, const FindFunction &find, const MakeFunction &make,
                             const std::string &abi)
    : _name(name)
{
    if (abi != SOAPY_SDR_ABI_VERSION) {
        SoapySDR_logf(SOAPY_SDR_ERROR, "module %s built for ABI %s, library is %s", name.c_str(), abi.c_str(),
                      SOAPY_SDR_ABI_VERSION);
        _name.clear();
        return;
    }
    table()[name] = Entry{find, make};
}

SoapySDR::Registry::~Registry(void)
{
    if (!_name.empty()) table().erase(_name);
}

SoapySDR::FindFunctions SoapySDR::Registry::listFindFunctions(void)
{
    FindFunctions f;
    for (const auto &kv : table()) f[kv.first] = kv.second.find;
    return f;
}

SoapySDR::MakeFunctions SoapySDR::Registry::listMakeFunctions(void)
{
    MakeFunctions f;
    for (const auto &kv : table()) f[kv.first] = kv.second.make;
    return f;
}

// ---- Device factory + defaults ----------------------------------------------
SoapySDR::Device::~Device(void) {}

SoapySDR::KwargsList SoapySDR::Device::enumerate(const Kwargs &args)
{
    KwargsList out;
    for (const auto &kv : table()) {
        if (args.count("driver") && args.at("driver") != kv.first) continue;
        for (auto found : kv.second.find(args)) {
            found["driver"] = kv.first;
            out.push_back(found);
        }
    }
    return out;
}

SoapySDR::KwargsList SoapySDR::Device::enumerate(const std::string &args) { return enumerate(KwargsFromString(args)); }

SoapySDR::Device *SoapySDR::Device::make(const Kwargs &args)
{
    const KwargsList found = enumerate(args);
    if (found.empty()) throw std::runtime_error("SoapySDR::Device::make() no match");
    Kwargs merged = found.front();
    for (const auto &kv : args) merged[kv.first] = kv.second;
    return table().at(merged.at("driver")).make(merged);
}

SoapySDR::Device *SoapySDR::Device::make(const std::string &args) { return make(KwargsFromString(args)); }

void SoapySDR::Device::unmake(Device *device) { delete device; }

std::string SoapySDR::Device::getDriverKey(void) const { return ""; }
std::string SoapySDR::Device::getHardwareKey(void) const { return ""; }
SoapySDR::Kwargs SoapySDR::Device::getHardwareInfo(void) const { return Kwargs(); }
size_t SoapySDR::Device::getNumChannels(const int) const { return 0; }
std::vector<std::string> SoapySDR::Device::getStreamFormats(const int, const size_t) const { return {}; }
std::string SoapySDR::Device::getNativeStreamFormat(const int, const size_t, double &fullScale) const
{
    fullScale = double(1 << 15);
    return SOAPY_SDR_CS16;
}
SoapySDR::Stream *SoapySDR::Device::setupStream(const int, const std::string &, const std::vector<size_t> &,
                                                const Kwargs &)
{
    return nullptr;
}
void SoapySDR::Device::closeStream(Stream *) {}
size_t SoapySDR::Device::getStreamMTU(Stream *) const { return 1024; }
int SoapySDR::Device::activateStream(Stream *, const int flags, const long long, const size_t)
{
    return (flags == 0) ? 0 : SOAPY_SDR_NOT_SUPPORTED;
}
int SoapySDR::Device::deactivateStream(Stream *, const int flags, const long long)
{
    return (flags == 0) ? 0 : SOAPY_SDR_NOT_SUPPORTED;
}
int SoapySDR::Device::readStream(Stream *, void *const *, const size_t, int &, long long &, const long)
{
    return SOAPY_SDR_NOT_SUPPORTED;
}
int SoapySDR::Device::writeStream(Stream *, const void *const *, const size_t, int &, const long long, const long)
{
    return SOAPY_SDR_NOT_SUPPORTED;
}
std::vector<std::string> SoapySDR::Device::listAntennas(const int, const size_t) const { return {}; }
void SoapySDR::Device::setAntenna(const int, const size_t, const std::string &) {}
std::string SoapySDR::Device::getAntenna(const int, const size_t) const { return ""; }
std::vector<std::string> SoapySDR::Device::listGains(const int, const size_t) const { return {}; }
void SoapySDR::Device::setGain(const int, const size_t, const double) {}
void SoapySDR::Device::setGain(const int, const size_t, const std::string &, const double) {}
double SoapySDR::Device::getGain(const int, const size_t) const { return 0.0; }
double SoapySDR::Device::getGain(const int, const size_t, const std::string &) const { return 0.0; }
SoapySDR::Range SoapySDR::Device::getGainRange(const int, const size_t) const { return Range(0.0, 0.0); }
SoapySDR::Range SoapySDR::Device::getGainRange(const int, const size_t, const std::string &) const
{
    return Range(0.0, 0.0);
}
void SoapySDR::Device::setFrequency(const int, const size_t, const double, const Kwargs &) {}
double SoapySDR::Device::getFrequency(const int, const size_t) const { return 0.0; }
SoapySDR::RangeList SoapySDR::Device::getFrequencyRange(const int, const size_t) const { return {}; }
void SoapySDR::Device::setSampleRate(const int, const size_t, const double) {}
double SoapySDR::Device::getSampleRate(const int, const size_t) const { return 0.0; }
std::vector<double> SoapySDR::Device::listSampleRates(const int, const size_t) const { return {}; }
SoapySDR::RangeList SoapySDR::Device::getSampleRateRange(const int, const size_t) const { return {}; }
bool SoapySDR::Device::hasHardwareTime(const std::string &) const { return false; }
long long SoapySDR::Device::getHardwareTime(const std::string &) const { return 0; }
void SoapySDR::Device::setHardwareTime(const long long, const std::string &) {}
void SoapySDR::Device::writeSetting(const std::string &, const std::string &) {}
std::string SoapySDR::Device::readSetting(const std::string &) const { return ""; }
void SoapySDR::Device::writeRegister(const std::string &, const unsigned, const unsigned) {}
unsigned SoapySDR::Device::readRegister(const std::string &, const unsigned) const { return 0; }
void SoapySDR::Device::writeRegisters(const std::string &, const unsigned, const std::vector<unsigned> &) {}
std::vector<unsigned> SoapySDR::Device::readRegisters(const std::string &, const unsigned, const size_t length) const
{
    return std::vector<unsigned>(length, 0);
}

// ---- Logger -----------------------------------------------------------------
extern "C" {

void SoapySDR_log(const SoapySDRLogLevel logLevel, const char *message)
{
    if (logLevel > g_level && logLevel != SOAPY_SDR_SSI) return;
    std::lock_guard<std::mutex> lock(log_mutex());
    g_handler(logLevel, message);
}

void SoapySDR_vlogf(const SoapySDRLogLevel logLevel, const char *format, va_list argList)
{
    if (logLevel > g_level && logLevel != SOAPY_SDR_SSI) return;
    char buf[1024];
    vsnprintf(buf, sizeof(buf), format, argList);
    SoapySDR_log(logLevel, buf);
}

void SoapySDR_logf(const SoapySDRLogLevel logLevel, const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    SoapySDR_vlogf(logLevel, format, ap);
    va_end(ap);
}

void SoapySDR_registerLogHandler(const SoapySDRLogHandler handler)
{
    std::lock_guard<std::mutex> lock(log_mutex());
    g_handler = handler ? handler : default_handler;
}

void SoapySDR_setLogLevel(const SoapySDRLogLevel logLevel) { g_level = logLevel; }
SoapySDRLogLevel SoapySDR_getLogLevel(void) { return g_level; }

// ---- Time: whole seconds in integers, remainder in double -------------------
long long SoapySDR_ticksToTimeNs(const long long ticks, const double rate)
{
    const long long ratell = (long long)rate;
    const long long full = ticks / ratell;
    const long long err = ticks - full * ratell;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = (((double)err - part) * 1e9) / rate;
    return full * 1000000000LL + std::llround(frac);
}

long long SoapySDR_timeNsToTicks(const long long timeNs, const double rate)
{
    const long long ratell = (long long)rate;
    const long long full = timeNs / 1000000000LL;
    const long long err = timeNs - full * 1000000000LL;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = part + ((double)err * rate) / 1e9;
    return full * ratell + std::llround(frac);
}

}  // extern "C"

void SoapySDR::log(const LogLevel logLevel, const std::string &message) { SoapySDR_log(logLevel, message.c_str()); }
void SoapySDR::vlogf(const SoapySDRLogLevel logLevel, const char *format, va_list argList)
{
    SoapySDR_vlogf(logLevel, format, argList);
}
void SoapySDR::logf(const SoapySDRLogLevel logLevel, const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    SoapySDR_vlogf(logLevel, format, ap);
    va_end(ap);
}
void SoapySDR::registerLogHandler(const LogHandler &handler) { SoapySDR_registerLogHandler(handler); }
void SoapySDR::setLogLevel(const LogLevel logLevel) { SoapySDR_setLogLevel(logLevel); }
