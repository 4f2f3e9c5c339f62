// Minimal API-compatible subset of <SoapySDR/Device.hpp> (see Constants.h):
// the virtuals that tejeez/sxxcvr's SoapySX overrides (SoapySX.cpp:740-1623),
// with the signatures of the public SoapySDR 0.8 API.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "Types.hpp"

namespace SoapySDR {

class Stream;

class Device {
public:
    virtual ~Device(void);

    static KwargsList enumerate(const Kwargs &args = Kwargs());
    static KwargsList enumerate(const std::string &args);
    static Device *make(const Kwargs &args = Kwargs());
    static Device *make(const std::string &args);
    static void unmake(Device *device);

    // identification
    virtual std::string getDriverKey(void) const;
    virtual std::string getHardwareKey(void) const;
    virtual Kwargs getHardwareInfo(void) const;

    // channels
    virtual size_t getNumChannels(const int direction) const;

    // streams
    virtual std::vector<std::string> getStreamFormats(const int direction, const size_t channel) const;
    virtual std::string getNativeStreamFormat(const int direction, const size_t channel, double &fullScale) const;
    virtual Stream *setupStream(const int direction, const std::string &format,
                                const std::vector<size_t> &channels = std::vector<size_t>(),
                                const Kwargs &args = Kwargs());
    virtual void closeStream(Stream *stream);
    virtual size_t getStreamMTU(Stream *stream) const;
    virtual int activateStream(Stream *stream, const int flags = 0, const long long timeNs = 0,
                               const size_t numElems = 0);
    virtual int deactivateStream(Stream *stream, const int flags = 0, const long long timeNs = 0);
    virtual int readStream(Stream *stream, void *const *buffs, const size_t numElems, int &flags,
                           long long &timeNs, const long timeoutUs = 100000);
    virtual int writeStream(Stream *stream, const void *const *buffs, const size_t numElems, int &flags,
                            const long long timeNs = 0, const long timeoutUs = 100000);

    // antennas
    virtual std::vector<std::string> listAntennas(const int direction, const size_t channel) const;
    virtual void setAntenna(const int direction, const size_t channel, const std::string &name);
    virtual std::string getAntenna(const int direction, const size_t channel) const;

    // gains
    virtual std::vector<std::string> listGains(const int direction, const size_t channel) const;
    virtual void setGain(const int direction, const size_t channel, const double value);
    virtual void setGain(const int direction, const size_t channel, const std::string &name, const double value);
    virtual double getGain(const int direction, const size_t channel) const;
    virtual double getGain(const int direction, const size_t channel, const std::string &name) const;
    virtual Range getGainRange(const int direction, const size_t channel) const;
    virtual Range getGainRange(const int direction, const size_t channel, const std::string &name) const;

    // frequency
    virtual void setFrequency(const int direction, const size_t channel, const double frequency,
                              const Kwargs &args = Kwargs());
    virtual double getFrequency(const int direction, const size_t channel) const;
    virtual RangeList getFrequencyRange(const int direction, const size_t channel) const;

    // sample rate
    virtual void setSampleRate(const int direction, const size_t channel, const double rate);
    virtual double getSampleRate(const int direction, const size_t channel) const;
    virtual std::vector<double> listSampleRates(const int direction, const size_t channel) const;
    virtual RangeList getSampleRateRange(const int direction, const size_t channel) const;

    // time
    virtual bool hasHardwareTime(const std::string &what = "") const;
    virtual long long getHardwareTime(const std::string &what = "") const;
    virtual void setHardwareTime(const long long timeNs, const std::string &what = "");

    // settings
    virtual void writeSetting(const std::string &key, const std::string &value);
    virtual std::string readSetting(const std::string &key) const;

    // registers
    virtual void writeRegister(const std::string &name, const unsigned addr, const unsigned value);
    virtual unsigned readRegister(const std::string &name, const unsigned addr) const;
    virtual void writeRegisters(const std::string &name, const unsigned addr, const std::vector<unsigned> &value);
    virtual std::vector<unsigned> readRegisters(const std::string &name, const unsigned addr,
                                                const size_t length) const;
};

}  // namespace SoapySDR
