// Flat C view (include/sx_device.h) of the SoapySDR Device registered as "sx".
#include "../../include/sx_device.h"

#include <SoapySDR/Device.hpp>
#include <SoapySDR/Logger.hpp>
#include <SoapySDR/Time.hpp>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

extern "C" int sx_device_internal_tx_capture(SoapySDR::Device *dev, long long dac_pos, size_t n, float *dst);

namespace {

thread_local std::string g_error;
std::mutex g_log_mutex;
std::string g_log;

void capture_log(const SoapySDRLogLevel level, const char *message)
{
    std::lock_guard<std::mutex> lock(g_log_mutex);
    static const char *names[] = {"", "FATAL", "CRITICAL", "ERROR", "WARNING", "NOTICE", "INFO", "DEBUG", "TRACE", "SSI"};
    g_log += "[";
    g_log += names[level <= 9 ? level : 0];
    g_log += "] ";
    g_log += message;
    g_log += "\n";
    if (g_log.size() > (1u << 20)) g_log.erase(0, g_log.size() - (1u << 19));
}

struct LogInit {
    LogInit() { SoapySDR_registerLogHandler(capture_log); }
} g_log_init;

SoapySDR::Device *D(sx_device *d) { return reinterpret_cast<SoapySDR::Device *>(d); }
SoapySDR::Stream *S(sx_stream *s) { return reinterpret_cast<SoapySDR::Stream *>(s); }

int copy_out(const std::string &s, char *out, size_t cap)
{
    if (!out || cap == 0) return (int)s.size();
    snprintf(out, cap, "%s", s.c_str());
    return (int)s.size();
}

}  // namespace

#define SX_TRY(default_ret, body)                 \
    try {                                         \
        body                                      \
    } catch (const std::exception &e) {           \
        g_error = e.what();                       \
        return default_ret;                       \
    } catch (...) {                               \
        g_error = "unknown exception";            \
        return default_ret;                       \
    }

extern "C" {

const char *sx_device_last_error(void) { return g_error.c_str(); }

int sx_device_enumerate(const char *args, char *out, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const auto found = SoapySDR::Device::enumerate(std::string(args ? args : ""));
        std::string s;
        for (const auto &kw : found) {
            if (!s.empty()) s += ";";
            s += SoapySDR::KwargsToString(kw);
        }
        copy_out(s, out, cap);
        return (int)found.size();
    })
}

sx_device *sx_device_make(const char *args)
{
    SX_TRY(nullptr, { return reinterpret_cast<sx_device *>(SoapySDR::Device::make(std::string(args ? args : ""))); })
}

int sx_device_unmake(sx_device *dev)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        SoapySDR::Device::unmake(D(dev));
        return 0;
    })
}

sx_stream *sx_device_setup_stream(sx_device *dev, int direction, const char *format, const size_t *channels,
                                  size_t num_chans, const char *args)
{
    SX_TRY(nullptr, {
        std::vector<size_t> ch;
        if (channels) ch.assign(channels, channels + num_chans);
        return reinterpret_cast<sx_stream *>(
            D(dev)->setupStream(direction, format ? format : "", ch, SoapySDR::KwargsFromString(args ? args : "")));
    })
}

int sx_device_close_stream(sx_device *dev, sx_stream *stream)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->closeStream(S(stream));
        return 0;
    })
}

long sx_device_get_stream_mtu(sx_device *dev, sx_stream *stream)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return (long)D(dev)->getStreamMTU(S(stream)); })
}

int sx_device_activate_stream(sx_device *dev, sx_stream *stream, int flags, long long time_ns, size_t num_elems)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return D(dev)->activateStream(S(stream), flags, time_ns, num_elems); })
}

int sx_device_deactivate_stream(sx_device *dev, sx_stream *stream, int flags, long long time_ns)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return D(dev)->deactivateStream(S(stream), flags, time_ns); })
}

int sx_device_read_stream(sx_device *dev, sx_stream *stream, void *const *buffs, size_t num_elems, int *flags,
                          long long *time_ns, long timeout_us)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return D(dev)->readStream(S(stream), buffs, num_elems, *flags, *time_ns, timeout_us); })
}

int sx_device_write_stream(sx_device *dev, sx_stream *stream, const void *const *buffs, size_t num_elems, int *flags,
                           long long time_ns, long timeout_us)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return D(dev)->writeStream(S(stream), buffs, num_elems, *flags, time_ns, timeout_us); })
}

int sx_device_has_hardware_time(sx_device *dev, const char *what)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return D(dev)->hasHardwareTime(what ? what : "") ? 1 : 0; })
}

int sx_device_get_hardware_time(sx_device *dev, const char *what, long long *time_ns)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        *time_ns = D(dev)->getHardwareTime(what ? what : "");
        return 0;
    })
}

int sx_device_list_sample_rates(sx_device *dev, int direction, size_t channel, double *rates, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const auto r = D(dev)->listSampleRates(direction, channel);
        for (size_t i = 0; i < r.size() && i < cap; ++i) rates[i] = r[i];
        return (int)r.size();
    })
}

int sx_device_set_sample_rate(sx_device *dev, int direction, size_t channel, double rate)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->setSampleRate(direction, channel, rate);
        return 0;
    })
}

double sx_device_get_sample_rate(sx_device *dev, int direction, size_t channel)
{
    SX_TRY(-1.0, { return D(dev)->getSampleRate(direction, channel); })
}

int sx_device_get_num_channels(sx_device *dev, int direction)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return (int)D(dev)->getNumChannels(direction); })
}

int sx_device_get_info(sx_device *dev, const char *what, int direction, char *out, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const std::string w(what ? what : "");
        std::string s;
        if (w == "driver_key") s = D(dev)->getDriverKey();
        else if (w == "hardware_key") s = D(dev)->getHardwareKey();
        else if (w == "hardware_info") s = SoapySDR::KwargsToString(D(dev)->getHardwareInfo());
        else if (w == "stream_formats") {
            for (const auto &f : D(dev)->getStreamFormats(direction, 0)) s += (s.empty() ? "" : ",") + f;
        } else if (w == "native_stream_format") {
            double fs = 0.0;
            s = D(dev)->getNativeStreamFormat(direction, 0, fs);
            s += "," + std::to_string(fs);
        } else {
            g_error = "unknown info key";
            return SX_DEVICE_EXCEPTION;
        }
        return copy_out(s, out, cap);
    })
}

int sx_device_set_frequency(sx_device *dev, int direction, size_t channel, double hz)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->setFrequency(direction, channel, hz, SoapySDR::Kwargs());
        return 0;
    })
}
double sx_device_get_frequency(sx_device *dev, int direction, size_t channel)
{
    SX_TRY(-1.0, { return D(dev)->getFrequency(direction, channel); })
}
int sx_device_set_gain(sx_device *dev, int direction, size_t channel, double db)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->setGain(direction, channel, db);
        return 0;
    })
}
double sx_device_get_gain(sx_device *dev, int direction, size_t channel)
{
    SX_TRY(-1.0, { return D(dev)->getGain(direction, channel); })
}
int sx_device_set_antenna(sx_device *dev, int direction, size_t channel, const char *name)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->setAntenna(direction, channel, name ? name : "");
        return 0;
    })
}
int sx_device_get_antenna(sx_device *dev, int direction, size_t channel, char *out, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return copy_out(D(dev)->getAntenna(direction, channel), out, cap); })
}

int sx_device_set_gain_element(sx_device *dev, int direction, size_t channel, const char *name, double db)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->setGain(direction, channel, std::string(name ? name : ""), db);
        return 0;
    })
}

int sx_device_get_gain_range(sx_device *dev, int direction, size_t channel, const char *name, double out[3])
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const SoapySDR::Range r = (name && name[0]) ? D(dev)->getGainRange(direction, channel, name)
                                                    : D(dev)->getGainRange(direction, channel);
        out[0] = r.minimum();
        out[1] = r.maximum();
        out[2] = r.step();
        return 0;
    })
}

double sx_device_get_gain_element(sx_device *dev, int direction, size_t channel, const char *name)
{
    SX_TRY(-1.0, { return D(dev)->getGain(direction, channel, std::string(name ? name : "")); })
}

int sx_device_list(sx_device *dev, const char *what, int direction, char *out, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const std::string w(what ? what : "");
        std::vector<std::string> items;
        if (w == "gains") items = D(dev)->listGains(direction, 0);
        else if (w == "antennas") items = D(dev)->listAntennas(direction, 0);
        else {
            g_error = "unknown list";
            return SX_DEVICE_EXCEPTION;
        }
        std::string s;
        for (const auto &i : items) s += (s.empty() ? "" : ",") + i;
        copy_out(s, out, cap);
        return (int)items.size();
    })
}

int sx_device_write_registers(sx_device *dev, const char *name, unsigned addr, const unsigned *values, size_t n)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->writeRegisters(name ? name : "", addr, std::vector<unsigned>(values, values + n));
        return 0;
    })
}

int sx_device_read_registers(sx_device *dev, const char *name, unsigned addr, unsigned *values, size_t n)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        const auto r = D(dev)->readRegisters(name ? name : "", addr, n);
        for (size_t i = 0; i < n && i < r.size(); ++i) values[i] = r[i];
        return (int)r.size();
    })
}

int sx_device_write_setting(sx_device *dev, const char *key, const char *value)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        D(dev)->writeSetting(key ? key : "", value ? value : "");
        return 0;
    })
}

int sx_device_read_setting(sx_device *dev, const char *key, char *out, size_t cap)
{
    SX_TRY(SX_DEVICE_EXCEPTION, { return copy_out(D(dev)->readSetting(key ? key : ""), out, cap); })
}

int sx_device_tx_capture(sx_device *dev, long long dac_pos, size_t n, float *dst)
{
    SX_TRY(SX_DEVICE_EXCEPTION, {
        if (sx_device_internal_tx_capture(D(dev), dac_pos, n, dst) != 0) {
            g_error = "not an sx device";
            return SX_DEVICE_EXCEPTION;
        }
        return 0;
    })
}

long long sx_ticks_to_time_ns(long long ticks, double rate) { return SoapySDR::ticksToTimeNs(ticks, rate); }
long long sx_time_ns_to_ticks(long long time_ns, double rate) { return SoapySDR::timeNsToTicks(time_ns, rate); }

int sx_device_set_log_level(int level)
{
    SoapySDR_setLogLevel((SoapySDRLogLevel)level);
    return 0;
}

int sx_device_drain_log(char *out, size_t cap)
{
    std::lock_guard<std::mutex> lock(g_log_mutex);
    const int n = copy_out(g_log, out, cap);
    g_log.clear();
    return n;
}

}  // extern "C"
