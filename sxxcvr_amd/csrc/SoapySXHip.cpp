// SoapySDR Device module "sx" for MI355X: the Device/Stream plugin surface of
// tejeez/sxxcvr (class SoapySX, SoapySX/SoapySX.cpp:524-1656) with the
// SX1255-over-I2S/ALSA sample feed replaced by a synthetic CF32 IQ source/sink
// and the chip's decimator/interpolator replaced by HIP kernels reached
// through the extern "C" shim of include/sxfir.h.
//
// Kept with the reference's meaning (file:line = SoapySX/SoapySX.cpp):
//   stream lifecycle + arguments threshold/link/period ........ :740-866
//   readStream: overrun skip, non-blocking clamp, timestamps ... :868-967
//   writeStream: timed placement, past-discard, underrun skip .. :969-1105
//   getHardwareTime / hasHardwareTime .......................... :1107-1139, :1618-1623
//   sample-rate table and validation ........................... :180-208, :1145-1219
//   formats / channels / keys / registration ................... :1567-1656
//   frequency / gain / antenna / raw register API .............. :1225-1561
//       kept as a register SHADOW: the same bit fields of the same SX1255
//       registers are updated and read back, nothing is sent over SPI
// Not carried over (no counterpart without the Raspberry Pi HAT): SPI, GPIO,
// chip reset, clock detection, HAT EEPROM.
//
// New device arguments (the reference ignores its device args, :711):
//   gpu=<n>            HIP device (default 0)
//   clock=virtual|wall sample clock model (default wall, like hardware)
//   master_clock=<Hz>  32e6 or 38.4e6 (default 38.4e6; the reference probes the PLL, :639-665)
//   decim=<D>          RX decimation done on the GPU (default 4)
//   interp=<L>         TX interpolation done on the GPU (default 4)
//   taps_per_phase=<n> filter length = n * ratio (default 32)
//   seed=<u64>         synthetic source seed (default 0x51255)
//   wire=cf32|s32      sample format on the synthetic chip side of the GPU kernels (default cf32);
//                      s32 = the reference's S32_LE I2S words: the RX decimator converts them on load
//                      (:103-112), the TX interpolator writes them with the keying bits (:116-137)
#include <SoapySDR/Device.hpp>
#include <SoapySDR/Logger.hpp>
#include <SoapySDR/Registry.hpp>
#include <SoapySDR/Time.hpp>

#include <algorithm>
#include <climits>
#include <cmath>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "GpuChains.hpp"
#include "Sx1255Shadow.hpp"
#include "StreamRules.hpp"
#include "SynthPcm.hpp"

const char *SoapySXHip_tag = "sx-mi355x";
// (the reference stamps its build with `git describe` / `git rev-parse` through CMake, SoapySX/CMakeLists.txt:18-43; here
// sxxcvr_amd/build.py and CMakeLists.txt pass -DSOAPYSX_COMMIT="<git rev-parse --short HEAD>"; a tree without git: "unknown")
#ifndef SOAPYSX_COMMIT
#define SOAPYSX_COMMIT "unknown"
#endif
const char *SoapySXHip_commit = SOAPYSX_COMMIT;

namespace {

struct sampleRateDiv {
    uint16_t div;    // ratio of master clock to sample rate
    uint8_t clkout;  // iism_clk_div, register 0x12 bits 3-0
    uint8_t mant;    // int_dec_mantisse, register 0x13 bit 7
    uint8_t m;       // int_dec_m_parameter, register 0x13 bit 6
    uint8_t n;       // int_dec_n_parameter, register 0x13 bits 5-3
};

// The six SX1255 I2S rates the reference supports and the register fields it programs for each (SoapySX.cpp:179-208;
// div = 8 * 3^m * 2^n).  tests/golden/rate_table.json holds the reference's own table; tests/test_gpu_device.py and
// tests/test_host_logic.py compare.
const sampleRateDiv sample_rates[] = {{1536, 6, 0, 1, 6}, {768, 4, 0, 1, 5}, {512, 3, 0, 0, 6},
                                      {256, 2, 0, 0, 5},  {128, 1, 0, 0, 4}, {64, 0, 0, 0, 3}};
const size_t N_SAMPLE_RATES = sizeof(sample_rates) / sizeof(sample_rates[0]);

int pcm_error_to_soapy_rx(int err) { return err == -EPIPE ? SOAPY_SDR_OVERFLOW : SOAPY_SDR_STREAM_ERROR; }
int pcm_error_to_soapy_tx(int err) { return err == -EPIPE ? SOAPY_SDR_UNDERFLOW : SOAPY_SDR_STREAM_ERROR; }

std::string arg(const SoapySDR::Kwargs &args, const char *key, const char *dflt)
{
    const auto it = args.find(key);
    return it == args.end() ? std::string(dflt) : it->second;
}


}  // namespace

class SoapySXHip : public SoapySDR::Device {
private:
    double masterClock;
    double sampleRate;
    mutable std::recursive_mutex reg_mutex;

    sx::SampleClock clock;
    sx::SynthPcm pcm_rx;
    sx::SynthPcm pcm_tx;
    float tx_threshold2;
    bool linked;

    int gpu;
    int decim, interp, taps_per_phase;
    bool decim_auto, interp_auto;   // ratio follows the sample rate like the chip's does: divider / 16
    uint32_t first_channel;
    int nchan;                  // channels per direction (device argument `channels`; the reference has 1, :1591-1595)
    int capture_channel;        // which TX channel tx_capture reads back (setting TX_CAPTURE_CHANNEL)
    uint64_t seed;
    bool wire_s32;
    std::unique_ptr<sx::RxChain> rx_chain;
    std::unique_ptr<sx::TxChain> tx_chain;
    sx::Sx1255Shadow chip;      // RF front-end control surface: register values only, no SPI
    std::vector<long> generic_logged;   // (ratio, taps per phase) shapes already reported by log_generic_once
    std::string pa_mode = "AUTO";   // writeSetting("PA", ON | OFF | AUTO), :1478-1491; AUTO = the state init leaves (:1487-1490)

    int64_t timestamp_to_samples(long long timestamp) const { return SoapySDR::timeNsToTicks(timestamp, sampleRate); }
    long long samples_to_timestamp(int64_t samples) const { return SoapySDR::ticksToTimeNs(samples, sampleRate); }

    // (Re)create the GPU chains for the current ratios.  Callers hold both stream mutexes (or are the
    // constructor).
    void rebuild_chains(bool rx, bool tx)
    {
        if (rx) {
            rx_chain.reset();      // frees the old chain's HBM before the new one allocates
            rx_chain.reset(new sx::RxChain(gpu, decim, taps_per_phase, seed, first_channel, nchan, wire_s32));
        }
        if (tx) {
            tx_chain.reset();
            // the synthetic sink retains the last ring_frames stream samples per channel at the DAC rate: 2^20 where
            // that stays within 256 MiB of HBM (a large write then is one or two passes, not one per ring wrap)
            size_t ring_frames = size_t(1) << 20;
            while (ring_frames > 65536 && 8 * ring_frames * (size_t)interp * (size_t)nchan > (size_t(256) << 20)) ring_frames /= 2;
            tx_chain.reset(new sx::TxChain(gpu, interp, taps_per_phase, ring_frames, nchan, wire_s32));
            tx_chain->set_threshold2(tx_threshold2);
        }
        // The LDS-tiled kernels exist for the ratios of the reference's rate table (4 .. 96, SoapySX.cpp:180-208) with
        // 32 taps per phase; any other `decim` / `interp` / `taps_per_phase` -- arguments this build added -- runs the
        // generic one-output-per-thread kernels: same bits, about 7 G wideband samples/s where the tiled ones run 500
        // (DESIGN.md section 5).  Said once per shape, not per call.
        if (rx) log_generic_once("RX decimator", decim, rx_chain->geometry());
        if (tx) log_generic_once("TX interpolator", interp, tx_chain->geometry());
    }

    void log_generic_once(const char *what, int ratio, const sxfir_geometry &g)
    {
        if (g.tiled) return;
        const long key = (long)ratio * 65536 + taps_per_phase;
        if (std::find(generic_logged.begin(), generic_logged.end(), key) != generic_logged.end()) return;
        generic_logged.push_back(key);
        SoapySDR_logf(SOAPY_SDR_INFO,
                      "%s: ratio %d with %d taps per phase has no LDS-tiled kernel (those cover ratios 4, 8, 16, 32, 48, 96 "
                      "with 32 taps per phase); it runs %s: the same results, roughly 80 times slower",
                      what, ratio, taps_per_phase, g.kernel);
    }

    void reset_streams()
    {
        pcm_rx.reset();
        pcm_tx.reset();
        if (rx_chain) rx_chain->reset();
        if (tx_chain) tx_chain->reset();
    }

public:
    SoapySXHip(const SoapySDR::Kwargs &args)
        : masterClock(std::stod(arg(args, "master_clock", "38.4e6"))),
          sampleRate(masterClock / 256.0),    // default after clock detection, :662
          clock(arg(args, "clock", "wall") == "virtual" ? sx::SampleClock::VIRTUAL : sx::SampleClock::WALL, sampleRate),
          pcm_rx("synth:rx", sx::SynthPcm::CAPTURE, &clock),
          pcm_tx("synth:tx", sx::SynthPcm::PLAYBACK, &clock),
          tx_threshold2(0.0f),
          linked(false),
          gpu(std::stoi(arg(args, "gpu", "0"))),
          decim(arg(args, "decim", "4") == "auto" ? 16 : std::stoi(arg(args, "decim", "4"))),
          interp(arg(args, "interp", "4") == "auto" ? 16 : std::stoi(arg(args, "interp", "4"))),
          taps_per_phase(std::stoi(arg(args, "taps_per_phase", "32"))),
          decim_auto(arg(args, "decim", "4") == "auto"),
          interp_auto(arg(args, "interp", "4") == "auto"),
          first_channel((uint32_t)std::stoul(arg(args, "first_channel", "0"))),
          nchan(std::stoi(arg(args, "channels", "1"))),
          capture_channel(0),
          seed(std::stoull(arg(args, "seed", "0x51255"), nullptr, 0)),
          wire_s32(arg(args, "wire", "cf32") == "s32"),
          chip(masterClock)
    {
        SoapySDR_logf(SOAPY_SDR_INFO, "Initializing SoapySX (MI355X synthetic-IQ build)");
        if (masterClock != 32.0e6 && masterClock != 38.4e6)
            throw std::runtime_error("master_clock must be 32e6 or 38.4e6");
        if (decim < 1 || interp < 1 || taps_per_phase < 1) throw std::runtime_error("bad decim/interp/taps_per_phase");
        if (nchan < 1 || nchan > 64) throw std::runtime_error("channels must be 1..64");
        int ndev = 0;
        if (sxfir_device_count(&ndev) != SXFIR_OK || ndev < 1)
            throw std::runtime_error(std::string("No MI355X visible: ") + sxfir_last_error());
        rebuild_chains(true, true);
    }

    ~SoapySXHip(void) { SoapySDR_logf(SOAPY_SDR_INFO, "Uninitializing SoapySX"); }

    /*******************************************************************
     * Sample streams
     ******************************************************************/

    SoapySDR::Stream *setupStream(const int direction, const std::string &format, const std::vector<size_t> &channels,
                                  const SoapySDR::Kwargs &args) override
    {
        // The reference's order (:750-764): both locks, format, running, already set up.
        std::scoped_lock lock(pcm_rx.mutex, pcm_tx.mutex);

        if (format != "CF32") throw std::runtime_error("Only CF32 format is currently supported");
        if (pcm_rx.state() == sx::SynthPcm::RUNNING || pcm_tx.state() == sx::SynthPcm::RUNNING)
            throw std::runtime_error("Streams can be setup only if none of the streams are running");

        auto *stream = direction == SOAPY_SDR_RX ? &pcm_rx : &pcm_tx;
        if (stream->setup_done) throw std::runtime_error("Stream has been setup already");

        // The reference has one channel and ignores the list (:747): so does a one-channel device here, whatever the
        // list holds.  Only with the build's own `channels=N` (N > 1) does the list mean something: a stream carries all
        // N channels, buffs[c] = channel c, and the list may be empty ("automatic") or name exactly those.
        if (nchan > 1 && !channels.empty()) {
            bool ok = channels.size() == (size_t)nchan;
            for (size_t i = 0; ok && i < channels.size(); ++i) ok = channels[i] == i;
            if (!ok) throw std::runtime_error("A stream carries all channels of the device: 0.." + std::to_string(nchan - 1));
        }

        if (stream->is_tx()) {
            const float tx_threshold_default = 1.0e-3;
            const float tx_threshold =
                (args.count("threshold") > 0) ? std::stof(args.at("threshold")) : tx_threshold_default;
            tx_threshold2 = tx_threshold * tx_threshold;
            tx_chain->set_threshold2(tx_threshold2);
        }

        const bool arg_link = (args.count("link") > 0 && args.at("link") == "1");
        stream->stream_mode = arg_link ? sx::STREAM_MODE_LINK : sx::STREAM_MODE_NORMAL;
        stream->configure(args.count("period") > 0 ? std::stoul(args.at("period")) : 0);
        stream->setup_done = 1;

        // RX and TX run from one sample clock once both exist (snd_pcm_link, :784-788)
        if ((!linked) && pcm_rx.setup_done && pcm_tx.setup_done) {
            SoapySDR_logf(SOAPY_SDR_DEBUG, "Linking streams");
            pcm_rx.link(&pcm_tx);
            linked = 1;
        }
        return reinterpret_cast<SoapySDR::Stream *>(stream);
    }

    void closeStream(SoapySDR::Stream *handle) override
    {
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        std::scoped_lock lock(stream->mutex);
        stream->setup_done = 0;
    }

    int activateStream(SoapySDR::Stream *handle, const int flags, const long long timeNs, const size_t numElems) override
    {
        (void)flags; (void)timeNs; (void)numElems;   // ignored like the reference, :810
        std::scoped_lock lock(pcm_rx.mutex, pcm_tx.mutex);
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        if (stream->activated) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "Stream was already activated");
            return SOAPY_SDR_STREAM_ERROR;
        }
        stream->activated = 1;
        if (stream->stream_mode == sx::STREAM_MODE_NORMAL) {
            if (stream->state() == sx::SynthPcm::PREPARED) stream->start();
        }
        return 0;
    }

    int deactivateStream(SoapySDR::Stream *handle, const int flags, const long long timeNs) override
    {
        (void)flags; (void)timeNs;
        std::scoped_lock lock(pcm_rx.mutex, pcm_tx.mutex);
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        if (!stream->activated) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "Stream was already deactivated");
            return SOAPY_SDR_STREAM_ERROR;
        }
        stream->activated = 0;
        // both inactive -> stop and rewind: position restarts at 0, :850-854
        if ((!pcm_rx.activated) && (!pcm_tx.activated)) {
            SoapySDR_logf(SOAPY_SDR_INFO, "Stopping and resetting streams");
            try {
                reset_streams();
            } catch (const std::exception &e) {
                SoapySDR_logf(SOAPY_SDR_ERROR, "reset: %s", e.what());
                return SOAPY_SDR_STREAM_ERROR;
            }
        }
        return 0;
    }

    size_t getStreamMTU(SoapySDR::Stream *handle) const override
    {
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        std::scoped_lock lock(stream->mutex);
        return stream->hwp_period_size;
    }

    int readStream(SoapySDR::Stream *handle, void *const *buffs, const size_t numElems, int &flags, long long &timeNs,
                   const long timeoutUs) override
    {
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        std::scoped_lock lock(stream->mutex);

        flags = 0;
        if (stream->is_tx()) throw std::runtime_error("Wrong direction");
        if (!stream->activated) return 0;

        int64_t pcm_avail = 0, pcm_delay = 0;
        const int avail_ret = stream->avail_delay(&pcm_avail, &pcm_delay);
        if (avail_ret < 0) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "rx avail_delay: %d", avail_ret);
            return pcm_error_to_soapy_rx(avail_ret);
        }
        SoapySDR_logf(SOAPY_SDR_DEBUG, "rx avail_delay: %d %ld %ld", avail_ret, (long)pcm_avail, (long)pcm_delay);

        // Overrun (the oldest samples were overwritten): jump ahead by the rule's catch-up distance, :910-927.
        if (const int64_t samples_to_skip = sx::rules::rx_overrun_skip(pcm_avail, stream->hwp_buffer_size, stream->hwp_period_size)) {
            const int64_t forwarded = stream->forward(samples_to_skip);
            if (forwarded < 0) {
                SoapySDR_logf(SOAPY_SDR_ERROR, "rx forward: %ld", (long)forwarded);
                return pcm_error_to_soapy_rx((int)forwarded);
            }
            stream->position += forwarded;
            pcm_avail -= forwarded;
            SoapySDR_logf(SOAPY_SDR_WARNING, "RX buffer overrun. Skipped %ld samples", (long)forwarded);
        }

        // blocking: the whole request; non-blocking: what is there now, :934-942
        uint64_t length = sx::rules::request_length((uint64_t)std::min(numElems, (size_t)INT_MAX), pcm_avail, timeoutUs);
        if (length == 0) return 0;

        int64_t first = 0;
        const int64_t samples_read = stream->begin_read((int64_t)length, &first);
        if (samples_read < 0) return pcm_error_to_soapy_rx((int)samples_read);

        timeNs = samples_to_timestamp(stream->position);   // :950
        flags |= SOAPY_SDR_HAS_TIME;
        stream->position += samples_read;

        try {
            // stream sample `first` == position before the read: the PCM frame
            // counter and `position` advance together
            rx_chain->produce(first, (size_t)samples_read, reinterpret_cast<float *const *>(buffs));
        } catch (const std::exception &e) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "rx chain: %s", e.what());
            return SOAPY_SDR_STREAM_ERROR;
        }
        return (int)samples_read;
    }

    int writeStream(SoapySDR::Stream *handle, const void *const *buffs, const size_t numElems, int &flags,
                    const long long timeNs, const long timeoutUs) override
    {
        auto *stream = reinterpret_cast<sx::SynthPcm *>(handle);
        std::scoped_lock lock(stream->mutex);

        if (!stream->is_tx()) throw std::runtime_error("Wrong direction");
        if (!stream->activated) return 0;

        int64_t pcm_avail = 0, pcm_delay = 0;
        const int avail_ret = stream->avail_delay(&pcm_avail, &pcm_delay);
        if (avail_ret < 0) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "tx avail_delay: %d", avail_ret);
            return pcm_error_to_soapy_tx(avail_ret);
        }
        SoapySDR_logf(SOAPY_SDR_DEBUG, "tx avail_delay: %d %ld %ld", avail_ret, (long)pcm_avail, (long)pcm_delay);

        uint64_t length = (uint64_t)std::min(numElems, (size_t)INT_MAX);
        // where the block lands: at its timestamp, behind the previous block, or past an underrun, :1000-1038
        const bool timed = (flags & SOAPY_SDR_HAS_TIME) != 0;
        const sx::rules::TxPlacement place = sx::rules::tx_placement(stream->position, pcm_delay, stream->hwp_period_size, timed,
                                                                    timed ? timestamp_to_samples(timeNs) : 0);
        if (place.kind == sx::rules::TxPlacement::IN_THE_PAST) {
            // dropped, but reported as written (:1014-1022)
            SoapySDR_logf(SOAPY_SDR_WARNING, "Discarding TX %ld samples in the past", (long)place.late);
            return (int)length;
        }
        if (place.kind == sx::rules::TxPlacement::PAST_UNDERRUN)
            SoapySDR_logf(SOAPY_SDR_WARNING, "TX buffer underrun. Forwarding TX stream by %ld samples",
                          (long)(place.write_position - stream->position));
        const int64_t write_position = place.write_position;

        // forward to the write position; what is skipped plays as silence, :1043-1073
        int64_t posdiff = write_position - stream->position;
        while (posdiff > 0) {
            const int64_t forwardable = stream->forwardable();
            if (forwardable < 0) {
                SoapySDR_logf(SOAPY_SDR_ERROR, "tx forwardable: %ld", (long)forwardable);
                return pcm_error_to_soapy_tx((int)forwardable);
            }
            int64_t forwarded;
            if (posdiff < forwardable) {
                forwarded = stream->forward(posdiff);
            } else {
                forwarded = stream->forward(forwardable);
                if (stream->state() != sx::SynthPcm::RUNNING && forwarded == 0) {
                    // a stopped ring never drains; the reference would wait forever in snd_pcm_wait
                    SoapySDR_logf(SOAPY_SDR_ERROR, "tx forward: stream is not running");
                    return SOAPY_SDR_TIMEOUT;
                }
                stream->wait_for_space_or_data();
            }
            if (forwarded < 0) {
                SoapySDR_logf(SOAPY_SDR_ERROR, "tx forward: %ld", (long)forwarded);
                return pcm_error_to_soapy_tx((int)forwarded);
            }
            stream->position += forwarded;
            posdiff -= forwarded;
            pcm_avail -= forwarded;
        }

        length = sx::rules::request_length(length, pcm_avail, timeoutUs);   // non-blocking: what fits now, :1076-1085
        if (length == 0) return 0;

        int64_t first = 0;
        const int64_t samples_written = stream->begin_write((int64_t)length, &first);
        if (samples_written < 0) return pcm_error_to_soapy_tx((int)samples_written);
        try {
            // (transmitter keying of convert_tx_buffer, :132-133, is counted on the GPU as the block passes:
            // TxChain::keyed_samples)
            tx_chain->consume(first, (size_t)samples_written, reinterpret_cast<const float *const *>(buffs));
        } catch (const std::exception &e) {
            SoapySDR_logf(SOAPY_SDR_ERROR, "tx chain: %s", e.what());
            return SOAPY_SDR_STREAM_ERROR;
        }
        stream->position += samples_written;
        return (int)samples_written;
    }

    long long getHardwareTime(const std::string &what) const override
    {
        if (what == "") {
            // TX side on purpose: does not contend with an RX thread, :1110-1125
            auto *stream = const_cast<sx::SynthPcm *>(&pcm_tx);
            std::scoped_lock lock(stream->mutex);
            int64_t pcm_avail = 0, pcm_delay = 0;
            const int ret = stream->avail_delay(&pcm_avail, &pcm_delay);
            if (ret < 0) throw std::runtime_error("PCM error");
            return samples_to_timestamp(stream->position - pcm_delay);
        }
        throw std::runtime_error("Unsupported time");
    }

    bool hasHardwareTime(const std::string &what) const override { return what == ""; }

    /*******************************************************************
     * Sample rates
     ******************************************************************/

    std::vector<double> listSampleRates(const int direction, const size_t channel) const override
    {
        (void)direction; (void)channel;
        std::vector<double> sampleRates;
        for (size_t i = 0; i < N_SAMPLE_RATES; i++) sampleRates.push_back(masterClock / (double)sample_rates[i].div);
        return sampleRates;
    }

    SoapySDR::RangeList getSampleRateRange(const int direction, const size_t channel) const override
    {
        SoapySDR::RangeList ranges;
        for (const auto rate : listSampleRates(direction, channel)) ranges.push_back({rate, rate, 0});
        return ranges;
    }

    void setSampleRate(const int direction, const size_t channel, const double rate) override
    {
        (void)direction; (void)channel;
        std::scoped_lock lock(pcm_rx.mutex, pcm_tx.mutex, reg_mutex);
        if (rate != rate || rate <= 0) throw std::runtime_error("Sample rate must be positive");
        const double divider = round(masterClock / rate);
        const sampleRateDiv *row = nullptr;
        for (size_t i = 0; i < N_SAMPLE_RATES; i++) {
            if ((double)sample_rates[i].div == divider) { row = &sample_rates[i]; break; }
        }
        if (!row) throw std::runtime_error("Unsupported sample rate");
        // the register shadow reads back what the reference would have programmed (:1192-1208): the I2S clock divider and
        // the decimator's mantissa / m / n fields, RX and TX enabled again afterwards
        chip.put({0x12, 0, 4}, row->clkout);
        chip.put({0x13, 7, 1}, row->mant);
        chip.put({0x13, 6, 1}, row->m);
        chip.put({0x13, 3, 3}, row->n);
        chip.put({0x00, 1, 2}, 3);
        sampleRate = masterClock / divider;
        clock.set_rate(sampleRate);
        // decim=auto / interp=auto: the converters run at master clock / 16 and the ratio follows the rate,
        // as the SX1255's own decimator and interpolator do when the divider registers are programmed
        // (:1192-1208): 600 kS/s -> 4, 300 kS/s -> 8, 150 -> 16, 75 -> 32, 50 -> 48, 25 -> 96 (at 38.4 MHz)
        const int ratio = (int)divider / 16;
        const bool new_rx = decim_auto && ratio != decim, new_tx = interp_auto && ratio != interp;
        if (new_rx) decim = ratio;
        if (new_tx) interp = ratio;
        if (new_rx || new_tx) rebuild_chains(new_rx, new_tx);
    }

    double getSampleRate(const int direction, const size_t channel) const override
    {
        (void)direction; (void)channel;
        std::scoped_lock lock(reg_mutex);
        return sampleRate;
    }

    /*******************************************************************
     * RF front-end control surface (SoapySX.cpp:1225-1561) on the register
     * shadow: same registers and bit fields, no SPI traffic (Sx1255Shadow.hpp)
     ******************************************************************/

    void setFrequency(const int direction, const size_t, const double frequency, const SoapySDR::Kwargs &) override
    {
        std::scoped_lock lock(reg_mutex);
        chip.tune(direction == SOAPY_SDR_RX, frequency);
    }

    double getFrequency(const int direction, const size_t) const override
    {
        std::scoped_lock lock(reg_mutex);
        return chip.tuned(direction == SOAPY_SDR_RX);
    }

    std::vector<std::string> listGains(const int direction, const size_t) const override
    {
        std::vector<std::string> names;
        for (const auto &e : sx::Sx1255Shadow::elements())
            if (e.rx == (direction == SOAPY_SDR_RX)) names.push_back(e.name);
        return names;
    }

    SoapySDR::Range getGainRange(const int direction, const size_t, const std::string &name) const override
    {
        const auto *e = sx::Sx1255Shadow::find(direction == SOAPY_SDR_RX, name);
        return e ? SoapySDR::Range(e->lo, e->hi, e->step) : SoapySDR::Range(0, 0, 0);
    }

    // SoapySDR's default for the overall range: element ranges added up (RX 0..78 dB, TX 0..39 dB)
    SoapySDR::Range getGainRange(const int direction, const size_t channel) const override
    {
        double lo = 0.0, hi = 0.0;
        for (const auto &name : listGains(direction, channel)) {
            const SoapySDR::Range r = getGainRange(direction, channel, name);
            lo += r.minimum();
            hi += r.maximum();
        }
        return SoapySDR::Range(lo, hi);
    }

    void setGain(const int direction, const size_t, const std::string &name, const double value) override
    {
        std::scoped_lock lock(reg_mutex);
        chip.set_gain(direction == SOAPY_SDR_RX, name, value);
    }

    double getGain(const int direction, const size_t, const std::string &name) const override
    {
        std::scoped_lock lock(reg_mutex);
        return chip.gain(direction == SOAPY_SDR_RX, name);
    }

    void setGain(const int direction, const size_t, const double value) override
    {
        std::scoped_lock lock(reg_mutex);
        chip.set_overall_gain(direction == SOAPY_SDR_RX, value);
    }

    // SoapySDR's default for the overall gain: the sum of the elements
    double getGain(const int direction, const size_t channel) const override
    {
        double total = 0.0;
        for (const auto &name : listGains(direction, channel)) total += getGain(direction, channel, name);
        return total;
    }

    std::vector<std::string> listAntennas(const int direction, const size_t) const override
    {
        // digital loop-back ("DLB") can be selected but is not advertised, as in the reference (:1407-1408)
        if (direction == SOAPY_SDR_RX) return {"RX", "LB"};
        return {"TX", "NONE"};
    }

    void setAntenna(const int direction, const size_t, const std::string &name) override
    {
        std::scoped_lock lock(reg_mutex);
        chip.set_antenna(direction == SOAPY_SDR_RX, name);
    }

    std::string getAntenna(const int direction, const size_t) const override
    {
        std::scoped_lock lock(reg_mutex);
        return chip.antenna(direction == SOAPY_SDR_RX);
    }

    std::vector<unsigned> readRegisters(const std::string &, const unsigned addr, const size_t length) const override
    {
        std::scoped_lock lock(reg_mutex);
        return chip.read(addr, length);
    }

    unsigned readRegister(const std::string &name, const unsigned addr) const override { return readRegisters(name, addr, 1).at(0); }

    void writeRegisters(const std::string &, const unsigned addr, const std::vector<unsigned> &value) override
    {
        std::scoped_lock lock(reg_mutex);
        chip.write(addr, value);
    }

    void writeRegister(const std::string &name, const unsigned addr, const unsigned value) override
    {
        writeRegisters(name, addr, std::vector<unsigned>{value});
    }

    /*******************************************************************
     * Settings: the virtual sample clock and the synthetic sink are driven
     * and inspected here (new; the reference only has "PA", :1472-1493)
     ******************************************************************/

    void writeSetting(const std::string &key, const std::string &value) override
    {
        if (key == "CLOCK_ADVANCE") {
            clock.advance(std::stoll(value));
        } else if (key == "PA") {
            // the reference drives two GPIO lines for ON / OFF / AUTO and does nothing for any other value (:1478-1491);
            // no GPIO here: the mode is remembered (readSetting("PA"), a key of this build), other values change nothing
            if (value == "ON" || value == "OFF" || value == "AUTO") {
                std::scoped_lock lock(reg_mutex);
                pa_mode = value;
            }
        } else if (key == "TX_CAPTURE_CHANNEL") {
            const int c = std::stoi(value);
            if (c < 0 || c >= nchan) throw std::runtime_error("No such channel");
            std::scoped_lock lock(pcm_tx.mutex);               // txCapture reads it under the TX mutex
            capture_channel = c;
        }
        // any other key: ignored without a word, as the reference does (:1472-1493 has no else branch)
    }

    std::string readSetting(const std::string &key) const override
    {
        // Any thread may ask while RX and TX stream on theirs.  The positions are atomics (no queueing behind a blocking
        // call); everything that looks into a chain takes that stream's mutex, as the stream calls do (:878, :979);
        // the ratios change under all three mutexes (setSampleRate) and are read under the register one.
        if (key == "CLOCK_NOW") return std::to_string(clock.now());
        if (key == "RX_POSITION") return std::to_string(pcm_rx.position.load());
        if (key == "TX_POSITION") return std::to_string(pcm_tx.position.load());
        if (key == "TX_WRITTEN") {
            std::scoped_lock lock(const_cast<sx::SynthPcm &>(pcm_tx).mutex);
            return std::to_string(tx_chain->written());
        }
        if (key == "TX_PTT_SAMPLES") {
            std::scoped_lock lock(const_cast<sx::SynthPcm &>(pcm_tx).mutex);
            return std::to_string(tx_chain->keyed_samples());
        }
        if (key == "RX_DIRECT_SAMPLES") {
            std::scoped_lock lock(const_cast<sx::SynthPcm &>(pcm_rx).mutex);
            return std::to_string(rx_chain->direct_samples());
        }
        if (key == "TX_DIRECT_SAMPLES") {
            std::scoped_lock lock(const_cast<sx::SynthPcm &>(pcm_tx).mutex);
            return std::to_string(tx_chain->direct_samples());
        }
        if (key == "RX_DECIM" || key == "TX_INTERP" || key == "RX_NTAPS") {
            std::scoped_lock lock(reg_mutex);
            return std::to_string(key == "RX_DECIM" ? decim : key == "TX_INTERP" ? interp : decim * taps_per_phase);
        }
        if (key == "SEED") return std::to_string(seed);
        if (key == "TX_CAPTURE_CHANNEL") {
            std::scoped_lock lock(const_cast<sx::SynthPcm &>(pcm_tx).mutex);
            return std::to_string(capture_channel);
        }
        if (key == "PA") {
            std::scoped_lock lock(reg_mutex);
            return pa_mode;
        }
        // the reference does not override readSetting (:1495 "TODO"): SoapySDR's default answers "" for every key
        return "";
    }

    // Synthetic sink inspection (used by the C ABI, include/sx_device.h)
    void txCapture(long long dac_pos, size_t n, float *dst)
    {
        std::scoped_lock lock(pcm_tx.mutex);
        tx_chain->capture(dac_pos, n, dst, capture_channel);
    }

    /*******************************************************************
     * Other hardware and driver information
     ******************************************************************/

    std::string getDriverKey(void) const override { return "sx"; }
    std::string getHardwareKey(void) const override { return "sx"; }

    SoapySDR::Kwargs getHardwareInfo(void) const override
    {
        SoapySDR::Kwargs args;
        args["soapysx_tag"] = SoapySXHip_tag;
        args["soapysx_commit"] = SoapySXHip_commit;
        args["hardware_version"] = "unknown";       // no HAT EEPROM to read, :1582-1587
        char name[64] = "", arch[32] = "";
        int cus = 0;
        size_t hbm = 0;
        if (sxfir_device_info(gpu, name, arch, &cus, &hbm) == SXFIR_OK) {
            args["gpu_name"] = name;
            args["gpu_arch"] = arch;
            args["gpu_compute_units"] = std::to_string(cus);
        }
        return args;
    }

    size_t getNumChannels(const int direction) const override { (void)direction; return (size_t)nchan; }

    std::string getNativeStreamFormat(const int direction, const size_t channel, double &fullScale) const override
    {
        (void)direction; (void)channel;
        fullScale = 1.0;
        return "CF32";
    }

    std::vector<std::string> getStreamFormats(const int direction, const size_t channel) const override
    {
        (void)direction; (void)channel;
        return {"CF32"};
    }
};

/***********************************************************************
 * Find / make / register (SoapySX.cpp:1629-1656)
 **********************************************************************/
static SoapySDR::KwargsList findDevice(const SoapySDR::Kwargs &args)
{
    (void)args;
    SoapySDR::KwargsList devices;
    SoapySDR::Kwargs device;
    device["label"] = "sx";
    device["driver"] = "sx";
    devices.push_back(device);
    return devices;
}

static SoapySDR::Device *makeDevice(const SoapySDR::Kwargs &args)
{
    SoapySDR::logf(SOAPY_SDR_INFO, "SoapySX version %s %s", SoapySXHip_tag, SoapySXHip_commit);
    return new SoapySXHip(args);
}

static SoapySDR::Registry registerDevice("sx", &findDevice, &makeDevice, SOAPY_SDR_ABI_VERSION);

// Used by the C ABI wrapper (sx_device_capi.cpp) for the sink read-back.
extern "C" int sx_device_internal_tx_capture(SoapySDR::Device *dev, long long dac_pos, size_t n, float *dst)
{
    auto *d = dynamic_cast<SoapySXHip *>(dev);
    if (!d) return -1;
    d->txCapture(dac_pos, n, dst);
    return 0;
}
