// GPU sample chains behind the SoapySX stream calls.  They stand where the
// SX1255 silicon and its I2S link stand in the reference: the chip decimates
// the ADC stream before snd_pcm_readi sees it (SoapySX.cpp:948) and
// interpolates what snd_pcm_writei hands it (SoapySX.cpp:1093).
//
//   RxChain: synthetic wideband CF32 IQ source (counter based, resident in
//            HBM) -> polyphase FIR decimator (HIP) -> host buffer
//   TxChain: host buffer -> polyphase FIR interpolator (HIP) -> DAC-rate ring
//            in HBM (the synthetic sink)
//
// Host C++ only: every GPU operation goes through the extern "C" shim in
// include/sxfir.h; this file does not include HIP.
#pragma once

#include <sxfir.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace sx {

inline void gpu_check(int rc, const char *what)
{
    if (rc != SXFIR_OK) throw std::runtime_error(std::string(what) + ": " + sxfir_last_error());
}

class DeviceBuffer {
public:
    DeviceBuffer() : ptr_(nullptr), bytes_(0) {}
    ~DeviceBuffer() { release(); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void reserve(size_t bytes)
    {
        if (bytes <= bytes_) return;
        release();
        gpu_check(sxfir_malloc(&ptr_, bytes), "sxfir_malloc");
        bytes_ = bytes;
    }
    void release()
    {
        if (ptr_) sxfir_free(ptr_);
        ptr_ = nullptr;
        bytes_ = 0;
    }
    void *get() const { return ptr_; }
    char *at(size_t byte_offset) const { return static_cast<char *>(ptr_) + byte_offset; }

private:
    void *ptr_;
    size_t bytes_;
};

class RxChain {
public:
    // wire_s32: the synthetic ADC stream is S32_LE I2S words and the decimator converts them on
    // load (the reference's wire format, SoapySX.cpp:103-112); otherwise CF32 end to end.
    RxChain(int gpu, int decim, int taps_per_phase, uint64_t seed, uint32_t channel, bool wire_s32)
        : decim_(decim), ntaps_(decim * taps_per_phase), seed_(seed), channel_(channel),
          fmt_(wire_s32 ? SXFIR_S32 : SXFIR_CF32), plan_(nullptr), next_(-1)
    {
        std::vector<float> taps((size_t)ntaps_);
        gpu_check(sxfir_design_lowpass(ntaps_, decim_, 8.0, 1.0, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_DECIMATE, taps.data(), ntaps_, decim_, 1, fmt_, gpu), "sxfir_create(rx)");
    }
    ~RxChain() { sxfir_destroy(plan_); }
    RxChain(const RxChain &) = delete;
    RxChain &operator=(const RxChain &) = delete;

    int decim() const { return decim_; }
    int ntaps() const { return ntaps_; }

    void reset() { next_ = -1; }

    // Deliver decimated stream samples [pos, pos+n) (stream rate) to host
    // memory as interleaved CF32.  Sample `m` of the stream is
    // sum_k h[k] * source[m*decim - k]; the source restarts at index 0 whenever
    // the PCMs are reset (stream position 0).
    void produce(int64_t pos, size_t n, float *host_dst)
    {
        const size_t block = 1u << 16;                       // outputs per GPU pass
        in_.reserve(sizeof(float) * 2 * block * (size_t)decim_);
        out_.reserve(sizeof(float) * 2 * block);
        if (pos != next_) prime(pos);
        size_t done = 0;
        while (done < n) {
            const size_t m = std::min(block, n - done);
            run(pos + (int64_t)done, m);
            gpu_check(sxfir_memcpy_d2h(host_dst + 2 * done, out_.get(), sizeof(float) * 2 * m, nullptr),
                      "sxfir_memcpy_d2h");
            gpu_check(sxfir_stream_sync(nullptr), "sxfir_stream_sync");
            done += m;
        }
        next_ = pos + (int64_t)n;
    }

private:
    // consume source samples [pos*D, (pos+m)*D) -> outputs [pos, pos+m)
    void run(int64_t pos, size_t m)
    {
        size_t n_out = 0;
        gpu_check(sxfir_synth_fill(in_.get(), m * (size_t)decim_, 0, 1, seed_, channel_, pos * decim_, fmt_, nullptr),
                  "sxfir_synth_fill");
        gpu_check(sxfir_decimate(plan_, in_.get(), m * (size_t)decim_, 0, out_.get(), 0, &n_out, nullptr),
                  "sxfir_decimate");
        if (n_out != m) throw std::runtime_error("rx chain: decimator produced an unexpected block size");
    }

    // After a skip (overrun) or a restart the filter history is rebuilt from
    // the source: run the ntaps samples that precede `pos` through the plan.
    void prime(int64_t pos)
    {
        gpu_check(sxfir_reset(plan_, nullptr), "sxfir_reset");
        const int64_t warm = (ntaps_ + decim_ - 1) / decim_;     // outputs whose inputs cover ntaps samples
        const int64_t from = pos - warm;                          // may be negative: source index < 0 is zero
        in_.reserve(sizeof(float) * 2 * (size_t)(warm * decim_));
        out_.reserve(sizeof(float) * 2 * (size_t)warm);
        size_t n_out = 0;
        gpu_check(sxfir_synth_fill(in_.get(), (size_t)(warm * decim_), 0, 1, seed_, channel_, from * decim_, fmt_,
                                   nullptr),
                  "sxfir_synth_fill");
        gpu_check(sxfir_decimate(plan_, in_.get(), (size_t)(warm * decim_), 0, out_.get(), 0, &n_out, nullptr),
                  "sxfir_decimate(prime)");
    }

    int decim_, ntaps_;
    uint64_t seed_;
    uint32_t channel_;
    int fmt_;
    sxfir_plan *plan_;
    DeviceBuffer in_, out_;
    int64_t next_;
};

class TxChain {
public:
    // wire_s32: the DAC-rate sink holds S32_LE I2S words with the transmitter-keying bits
    // (convert_tx_buffer, SoapySX.cpp:116-137, fused into the interpolator's store).
    TxChain(int gpu, int interp, int taps_per_phase, size_t ring_frames, bool wire_s32)
        : interp_(interp), ntaps_(interp * taps_per_phase), plan_(nullptr), ring_len_(ring_frames * (size_t)interp),
          next_(0), written_(0)
    {
        std::vector<float> taps((size_t)ntaps_);
        // gain = interp: unity pass-band gain after zero stuffing
        gpu_check(sxfir_design_lowpass(ntaps_, interp_, 8.0, (double)interp_, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_INTERPOLATE, taps.data(), ntaps_, interp_, 1,
                               wire_s32 ? SXFIR_S32 : SXFIR_CF32, gpu),
                  "sxfir_create(tx)");
        ring_.reserve(sizeof(float) * 2 * ring_len_);
        zeros_.assign(2 * 4096, 0.0f);
    }
    ~TxChain() { sxfir_destroy(plan_); }
    TxChain(const TxChain &) = delete;
    TxChain &operator=(const TxChain &) = delete;

    int interp() const { return interp_; }
    int64_t written() const { return written_; }
    void set_threshold2(float thr2) { gpu_check(sxfir_set_tx_threshold(plan_, thr2), "sxfir_set_tx_threshold"); }

    void reset()
    {
        gpu_check(sxfir_reset(plan_, nullptr), "sxfir_reset");
        next_ = 0;
        written_ = 0;
    }

    // Stream samples [pos, pos+n) from host memory.  Positions the application
    // skipped (snd_pcm_forward: timed gaps, underrun recovery) are silence.
    void consume(int64_t pos, size_t n, const float *host_src)
    {
        if (pos < next_) throw std::runtime_error("tx chain: position moved backwards");
        silence(pos - next_);
        feed(host_src, n);
        written_ += (int64_t)n;
    }

    // Read back DAC-rate samples [dac_pos, dac_pos+n) from the sink ring (only
    // the most recent ring_len samples are retained).
    void capture(int64_t dac_pos, size_t n, float *host_dst)
    {
        const int64_t end = next_ * interp_;
        if (dac_pos < 0 || dac_pos + (int64_t)n > end || end - dac_pos > (int64_t)ring_len_)
            throw std::runtime_error("tx capture: range not held by the sink ring");
        size_t done = 0;
        while (done < n) {
            const size_t off = (size_t)((dac_pos + (int64_t)done) % (int64_t)ring_len_);
            const size_t m = std::min(n - done, ring_len_ - off);
            gpu_check(sxfir_memcpy_d2h(host_dst + 2 * done, ring_.at(sizeof(float) * 2 * off), sizeof(float) * 2 * m,
                                       nullptr),
                      "sxfir_memcpy_d2h");
            done += m;
        }
        gpu_check(sxfir_stream_sync(nullptr), "sxfir_stream_sync");
    }

private:
    void silence(int64_t gap)
    {
        while (gap > 0) {
            const size_t m = (size_t)std::min<int64_t>(gap, (int64_t)zeros_.size() / 2);
            feed(zeros_.data(), m);
            gap -= (int64_t)m;
        }
    }

    // n stream samples -> n*interp ring samples at ring position next_*interp
    void feed(const float *host_src, size_t n)
    {
        const size_t block = 1u << 15;
        in_.reserve(sizeof(float) * 2 * block);
        size_t done = 0;
        while (done < n) {
            size_t m = std::min(block, n - done);
            const size_t off = (size_t)((next_ * interp_) % (int64_t)ring_len_);
            m = std::min(m, (ring_len_ - off) / (size_t)interp_);    // do not wrap inside one pass
            size_t n_out = 0;
            gpu_check(sxfir_memcpy_h2d(in_.get(), host_src + 2 * done, sizeof(float) * 2 * m, nullptr),
                      "sxfir_memcpy_h2d");
            gpu_check(sxfir_interpolate(plan_, in_.get(), m, 0, ring_.at(sizeof(float) * 2 * off), 0, &n_out, nullptr),
                      "sxfir_interpolate");
            gpu_check(sxfir_stream_sync(nullptr), "sxfir_stream_sync");   // host_src may be reused by the caller
            next_ += (int64_t)m;
            done += m;
        }
    }

    int interp_, ntaps_;
    sxfir_plan *plan_;
    size_t ring_len_;
    DeviceBuffer in_, ring_;
    std::vector<float> zeros_;
    int64_t next_;        // stream samples consumed so far (written + silence)
    int64_t written_;     // stream samples that carried application data
};

}  // namespace sx
