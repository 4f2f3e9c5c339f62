// GPU sample chains behind the SoapySX stream calls.  They stand where the
// SX1255 silicon and its I2S link stand in the reference: the chip decimates
// the ADC stream before snd_pcm_readi sees it (SoapySX.cpp:948) and
// interpolates what snd_pcm_writei hands it (SoapySX.cpp:1093).
//
//   RxChain: synthetic wideband IQ source (counter based, resident in HBM)
//            -> polyphase FIR decimator (HIP), which stores straight into
//            pinned host staging (zero copy) -> caller
//   TxChain: caller -> pinned host staging, read in place by the polyphase
//            FIR interpolator (HIP) -> DAC-rate ring in HBM (the synthetic sink)
// The narrow side of either resampler is 1/ratio of the traffic, so it crosses
// PCIe inside the kernel and the wide side never leaves HBM.
//
// Both are batched and asynchronous, which is what a GPU behind a 256-sample
// API needs: the RX side produces the stream in batches of thousands of
// samples on its own HIP stream and always has the NEXT batch in flight
// while the host hands out the current one from pinned memory (the source is
// a pure function of the stream position, so reading ahead cannot go wrong;
// a jump of the position - overrun skip, reset - drops the batches and
// re-primes the filter history).  The TX side copies the caller's block
// into a pinned slot and returns; the H2D copy and the interpolator run
// behind it and are only waited for when the slot comes round again.
// N channels (device argument `channels`) ride through the same launches
// (blockIdx.y = channel), BASELINE config 4's 8-per-GPU layout.
//
// Host C++ only: every GPU operation goes through the extern "C" shim in
// include/sxfir.h; this file does not include HIP.
#pragma once

#include <sxfir.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
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

// page-locked host memory: the only kind hipMemcpyAsync overlaps with kernels
class PinnedBuffer {
public:
    PinnedBuffer() : ptr_(nullptr), bytes_(0) {}
    ~PinnedBuffer() { release(); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    void reserve(size_t bytes)
    {
        if (bytes <= bytes_) return;
        release();
        gpu_check(sxfir_host_alloc(&ptr_, bytes), "sxfir_host_alloc");
        bytes_ = bytes;
    }
    void release()
    {
        if (ptr_) sxfir_host_free(ptr_);
        ptr_ = nullptr;
        bytes_ = 0;
    }
    float *floats() const { return static_cast<float *>(ptr_); }

private:
    void *ptr_;
    size_t bytes_;
};

class GpuStream {
public:
    GpuStream() : st_(nullptr) { gpu_check(sxfir_stream_create(&st_), "sxfir_stream_create"); }
    ~GpuStream() { sxfir_stream_destroy(st_); }
    GpuStream(const GpuStream &) = delete;
    GpuStream &operator=(const GpuStream &) = delete;
    void *get() const { return st_; }
    void sync() const { gpu_check(sxfir_stream_sync(st_), "sxfir_stream_sync"); }

private:
    void *st_;
};

class RxChain {
public:
    static constexpr size_t kMinBatch = 4096;       // stream samples per channel and GPU pass
    static constexpr size_t kMaxBatch = 1u << 16;

    // wire_s32: the synthetic ADC stream is S32_LE I2S words and the decimator converts them on
    // load (the reference's wire format, SoapySX.cpp:103-112); otherwise CF32 end to end.
    RxChain(int gpu, int decim, int taps_per_phase, uint64_t seed, uint32_t first_channel, int nchan, bool wire_s32)
        : gpu_(gpu), decim_(decim), ntaps_(decim * taps_per_phase), nchan_(nchan), seed_(seed),
          first_channel_(first_channel), fmt_(wire_s32 ? SXFIR_S32 : SXFIR_CF32), plan_(nullptr), next_(-1), cur_(0),
          batch_(kMinBatch)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        stream_.reset(new GpuStream());
        std::vector<float> taps((size_t)ntaps_);
        gpu_check(sxfir_design_lowpass(ntaps_, decim_, 8.0, 1.0, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_DECIMATE, taps.data(), ntaps_, decim_, nchan_, fmt_, gpu_), "sxfir_create(rx)");
        in_.reserve(8 * kMaxBatch * (size_t)decim_ * (size_t)nchan_);
        for (int k = 0; k < 2; ++k) stage_[k].reserve(8 * kMaxBatch * (size_t)nchan_);
    }
    ~RxChain()
    {
        sxfir_stream_sync(stream_->get());
        sxfir_destroy(plan_);
    }
    RxChain(const RxChain &) = delete;
    RxChain &operator=(const RxChain &) = delete;

    int decim() const { return decim_; }
    int ntaps() const { return ntaps_; }
    int channels() const { return nchan_; }

    void reset()
    {
        next_ = -1;
        slot_[0].n = slot_[1].n = 0;
    }

    // Deliver decimated stream samples [pos, pos+n) of every channel (stream rate) to host memory as
    // interleaved CF32, dsts[c] for channel c.  Sample `m` of a stream is sum_k h[k] * source[m*decim - k];
    // the source restarts at index 0 whenever the PCMs are reset (stream position 0).
    void produce(int64_t pos, size_t n, float *const *dsts)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        if (pos != next_) {
            // a jump (overrun skip, restart): whatever was read ahead is for the wrong positions
            slot_[0].n = slot_[1].n = 0;
            prime(pos);
            batch_ = pick_batch(n);
            launch(cur_, pos, batch_);
            launch(cur_ ^ 1, pos + (int64_t)batch_, batch_);
        }
        size_t done = 0;
        while (done < n) {
            Slot &s = slot_[cur_];
            const int64_t p = pos + (int64_t)done;
            if (s.n == 0 || p >= s.pos + (int64_t)s.n) {
                // current batch used up: the one in flight becomes current ...
                const int64_t following = s.n ? s.pos + (int64_t)s.n : p;
                cur_ ^= 1;
                if (slot_[cur_].n == 0) launch(cur_, following, batch_);
                wait(cur_);
                // ... and the batch after it goes in flight while the host hands this one out
                batch_ = pick_batch(n);
                launch(cur_ ^ 1, slot_[cur_].pos + (int64_t)slot_[cur_].n, batch_);
                continue;
            }
            wait(cur_);
            const size_t off = (size_t)(p - s.pos);
            const size_t m = std::min(n - done, s.n - off);
            for (int c = 0; c < nchan_; ++c)
                std::memcpy(dsts[c] + 2 * done, stage_[cur_].floats() + 2 * ((size_t)c * s.n + off), 8 * m);
            done += m;
        }
        next_ = pos + (int64_t)n;
    }

private:
    struct Slot {
        int64_t pos = 0;
        size_t n = 0;         // 0 = nothing launched into this slot
        bool ready = false;   // the host may read the staged samples
    };

    size_t pick_batch(size_t request) const
    {
        size_t b = kMinBatch;
        while (b < 4 * request && b < kMaxBatch) b *= 2;
        return b;
    }

    // stream samples [pos, pos+m) of all channels -> staging slot k, asynchronously on the chain's stream
    void launch(int k, int64_t pos, size_t m)
    {
        size_t n_out = 0;
        void *st = stream_->get();
        gpu_check(sxfir_synth_fill(in_.get(), m * (size_t)decim_, m * (size_t)decim_, nchan_, seed_, first_channel_,
                                   pos * decim_, fmt_, st),
                  "sxfir_synth_fill");
        // the decimator stores straight into the pinned staging buffer (device-visible host memory): the
        // outputs are 1/decim of the traffic and cross PCIe as they are produced, no separate D2H copy
        gpu_check(sxfir_decimate(plan_, in_.get(), m * (size_t)decim_, m * (size_t)decim_, stage_[k].floats(), m, &n_out, st),
                  "sxfir_decimate");
        if (n_out != m) throw std::runtime_error("rx chain: decimator produced an unexpected block size");
        slot_[k].pos = pos;
        slot_[k].n = m;
        slot_[k].ready = false;
    }

    void wait(int k)
    {
        if (slot_[k].ready) return;
        stream_->sync();                 // one stream, in order: everything launched so far has landed
        slot_[0].ready = slot_[0].n != 0;
        slot_[1].ready = slot_[1].n != 0;
    }

    // After a skip (overrun) or a restart the filter history is rebuilt from
    // the source: run the ntaps samples that precede `pos` through the plan.
    void prime(int64_t pos)
    {
        void *st = stream_->get();
        gpu_check(sxfir_reset(plan_, st), "sxfir_reset");
        const int64_t warm = (ntaps_ + decim_ - 1) / decim_;     // outputs whose inputs cover ntaps samples
        scratch_.reserve(8 * (size_t)warm * (size_t)nchan_);
        const int64_t from = pos - warm;                          // may be negative: source index < 0 is zero
        size_t n_out = 0;
        gpu_check(sxfir_synth_fill(in_.get(), (size_t)(warm * decim_), (size_t)(warm * decim_), nchan_, seed_,
                                   first_channel_, from * decim_, fmt_, st),
                  "sxfir_synth_fill");
        gpu_check(sxfir_decimate(plan_, in_.get(), (size_t)(warm * decim_), (size_t)(warm * decim_), scratch_.get(),
                                 (size_t)warm, &n_out, st),
                  "sxfir_decimate(prime)");
    }

    int gpu_, decim_, ntaps_, nchan_;
    uint64_t seed_;
    uint32_t first_channel_;
    int fmt_;
    sxfir_plan *plan_;
    std::unique_ptr<GpuStream> stream_;
    DeviceBuffer in_, scratch_;      // wideband source block; outputs of the priming pass (discarded)
    PinnedBuffer stage_[2];
    Slot slot_[2];
    int64_t next_;
    int cur_;
    size_t batch_;
};

class TxChain {
public:
    static constexpr size_t kSlotFrames = 1u << 15;    // stream samples per channel and pinned slot
    static constexpr size_t kFlushFrames = 4096;       // small writes are gathered up to this before a GPU pass
    static constexpr int kSlots = 4;

    // wire_s32: the DAC-rate sink holds S32_LE I2S words with the transmitter-keying bits
    // (convert_tx_buffer, SoapySX.cpp:116-137, fused into the interpolator's store).
    TxChain(int gpu, int interp, int taps_per_phase, size_t ring_frames, int nchan, bool wire_s32)
        : gpu_(gpu), interp_(interp), ntaps_(interp * taps_per_phase), nchan_(nchan), plan_(nullptr),
          ring_len_(ring_frames * (size_t)interp), next_(0), accepted_(0), written_(0), slot_(0), pend_(0)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        stream_.reset(new GpuStream());
        std::vector<float> taps((size_t)ntaps_);
        // gain = interp: unity pass-band gain after zero stuffing
        gpu_check(sxfir_design_lowpass(ntaps_, interp_, 8.0, (double)interp_, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_INTERPOLATE, taps.data(), ntaps_, interp_, nchan_,
                               wire_s32 ? SXFIR_S32 : SXFIR_CF32, gpu_),
                  "sxfir_create(tx)");
        ring_.reserve(8 * ring_len_ * (size_t)nchan_);
        stage_.reserve(8 * kSlotFrames * (size_t)nchan_ * kSlots);
        for (int k = 0; k < kSlots; ++k) {
            busy_[k] = false;
            done_[k] = nullptr;
            gpu_check(sxfir_event_create(&done_[k]), "sxfir_event_create");
        }
    }
    ~TxChain()
    {
        sxfir_stream_sync(stream_->get());
        for (int k = 0; k < kSlots; ++k) sxfir_event_destroy(done_[k]);
        sxfir_destroy(plan_);
    }
    TxChain(const TxChain &) = delete;
    TxChain &operator=(const TxChain &) = delete;

    int interp() const { return interp_; }
    int channels() const { return nchan_; }
    int64_t written() const { return written_; }
    void set_threshold2(float thr2) { gpu_check(sxfir_set_tx_threshold(plan_, thr2), "sxfir_set_tx_threshold"); }

    void reset()
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        pend_ = 0;                  // what was not passed to the GPU yet is dropped with the rest of the sink
        drain();
        gpu_check(sxfir_reset(plan_, stream_->get()), "sxfir_reset");
        next_ = 0;
        accepted_ = 0;
        written_ = 0;
    }

    // Stream samples [pos, pos+n) of every channel from host memory (srcs[c]).  Positions the
    // application skipped (snd_pcm_forward: timed gaps, underrun recovery) are silence.
    void consume(int64_t pos, size_t n, const float *const *srcs)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        if (pos < accepted_) throw std::runtime_error("tx chain: position moved backwards");
        feed(nullptr, (size_t)(pos - accepted_));
        feed(srcs, n);
        written_ += (int64_t)n;
    }

    // Read back DAC-rate samples [dac_pos, dac_pos+n) of one channel from the sink ring (only the
    // most recent ring_len samples are retained).
    void capture(int64_t dac_pos, size_t n, float *host_dst, int channel = 0)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        flush();
        const int64_t end = next_ * interp_;
        if (channel < 0 || channel >= nchan_) throw std::runtime_error("tx capture: no such channel");
        if (dac_pos < 0 || dac_pos + (int64_t)n > end || end - dac_pos > (int64_t)ring_len_)
            throw std::runtime_error("tx capture: range not held by the sink ring");
        size_t done = 0;
        while (done < n) {
            const size_t off = (size_t)((dac_pos + (int64_t)done) % (int64_t)ring_len_);
            const size_t m = std::min(n - done, ring_len_ - off);
            gpu_check(sxfir_memcpy_d2h(host_dst + 2 * done, ring_.at(8 * ((size_t)channel * ring_len_ + off)), 8 * m,
                                       stream_->get()),
                      "sxfir_memcpy_d2h");
            done += m;
        }
        drain();
    }

private:
    void drain()
    {
        stream_->sync();
        for (int k = 0; k < kSlots; ++k) busy_[k] = false;
    }

    // n stream samples per channel (srcs == nullptr: silence) are appended to the current pinned slot
    // (channel c at c*kSlotFrames); the GPU pass over a slot is issued once kFlushFrames have gathered
    // or the slot is full, and is never waited for here (write-behind: the sink is only observable
    // through capture(), which flushes).
    void feed(const float *const *srcs, size_t n)
    {
        size_t done = 0;
        while (done < n) {
            if (pend_ == 0 && busy_[slot_]) {
                // the slot's last H2D may still be reading it: wait for THAT pass only, not for the newer ones
                gpu_check(sxfir_event_sync(done_[slot_]), "sxfir_event_sync");
                busy_[slot_] = false;
            }
            const size_t m = std::min(kSlotFrames - pend_, n - done);
            float *host = stage_.floats() + 2 * kSlotFrames * (size_t)nchan_ * (size_t)slot_;
            for (int c = 0; c < nchan_; ++c) {
                float *dst = host + 2 * ((size_t)c * kSlotFrames + pend_);
                if (srcs) std::memcpy(dst, srcs[c] + 2 * done, 8 * m);
                else std::memset(dst, 0, 8 * m);
            }
            pend_ += m;
            accepted_ += (int64_t)m;
            done += m;
            if (pend_ >= kFlushFrames) flush();
        }
    }

    // pass the gathered samples of the current slot through the interpolator into the sink ring at ring
    // position next_*interp (in two passes where the ring wraps), asynchronously
    void flush()
    {
        if (pend_ == 0) return;
        void *st = stream_->get();
        const size_t slot_off = kSlotFrames * (size_t)nchan_ * (size_t)slot_;
        // the interpolator reads its input straight from the pinned slot (device-visible host memory): the
        // input is 1/interp of the traffic and crosses PCIe as the kernel fetches it, no separate H2D copy
        const char *host = reinterpret_cast<const char *>(stage_.floats() + 2 * slot_off);
        size_t done = 0;
        while (done < pend_) {
            const size_t off = (size_t)((next_ * interp_) % (int64_t)ring_len_);
            const size_t m = std::min(pend_ - done, (ring_len_ - off) / (size_t)interp_);
            size_t n_out = 0;
            gpu_check(sxfir_interpolate(plan_, host + 8 * done, m, kSlotFrames, ring_.at(8 * off), ring_len_, &n_out, st),
                      "sxfir_interpolate");
            next_ += (int64_t)m;
            done += m;
        }
        gpu_check(sxfir_event_record(done_[slot_], st), "sxfir_event_record");
        busy_[slot_] = true;
        slot_ = (slot_ + 1) % kSlots;
        pend_ = 0;
    }

    int gpu_, interp_, ntaps_, nchan_;
    sxfir_plan *plan_;
    size_t ring_len_;
    std::unique_ptr<GpuStream> stream_;
    DeviceBuffer ring_;
    PinnedBuffer stage_;
    bool busy_[kSlots];
    void *done_[kSlots];  // recorded behind each slot's GPU pass
    int64_t next_;        // stream samples passed to the GPU so far (written + silence)
    int64_t accepted_;    // next_ + what is gathered in the current slot
    int64_t written_;     // stream samples that carried application data
    int slot_;
    size_t pend_;         // samples gathered in the current slot and not yet passed to the GPU
};

}  // namespace sx
