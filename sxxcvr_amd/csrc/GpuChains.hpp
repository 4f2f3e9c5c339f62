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
// The narrow side of either resampler is 1/ratio of the traffic and is the only one that crosses PCIe: inside
// the kernel for small passes (zero copy), as one DMA-engine copy next to the kernel for passes of a megabyte
// and more; the wide side never leaves HBM.  Page-locked caller memory is used as it is (no staging copy).
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
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace sx {

inline void gpu_check(int rc, const char *what)
{
    if (rc != SXFIR_OK) throw std::runtime_error(std::string(what) + ": " + sxfir_last_error());
}

// memcpy of large blocks by a few PERSISTENT threads: one core moves ~10 GB/s between pageable caller memory and
// pinned staging, the GS/s block sizes of the wideband configurations want more, and starting threads per call
// costs as much as copying a megabyte.  Four threads by default: 8 and 16 measured no faster on the boxes seen
// (tools/devpath_probe.py; SX_COPY_THREADS overrides).  One pool per chain (RX and TX run on different application threads); the
// workers are started by the first block that is large enough and sleep on a condition variable in between.
class CopyPool {
public:
    static constexpr size_t kParallelFrom = size_t(1) << 20;   // smaller blocks: plain memcpy on the calling thread

    explicit CopyPool(unsigned max_threads = 4) : stop_(false), remaining_(0)
    {
        unsigned n = std::thread::hardware_concurrency();
        // SX_COPY_THREADS: upper bound of the threads one large copy is split over (1 = never start any)
        if (const char *e = std::getenv("SX_COPY_THREADS")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v >= 1 && v <= 64) max_threads = (unsigned)v;
        }
        nthreads_ = n < 2 ? 1 : (n > max_threads ? max_threads : n);
        if (const char *e = std::getenv("SX_COPY_PARALLEL_FROM")) {
            const long long v = std::strtoll(e, nullptr, 10);
            if (v >= 65536) parallel_from_ = (size_t)v;
        }
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        work_.notify_all();
        for (auto &w : workers_) w.join();
    }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    unsigned threads() const { return nthreads_; }
    size_t parallel_copies() const { return parallel_copies_; }

    void copy(void *dst, const void *src, size_t bytes)
    {
        if (bytes < parallel_from_ || nthreads_ < 2) {
            std::memcpy(dst, src, bytes);
            return;
        }
        const size_t parts = std::min<size_t>(nthreads_, bytes / (parallel_from_ / 4));
        const size_t piece = ((bytes / parts) + 4095) & ~size_t(4095);
        if (workers_.empty()) {
            jobs_.resize(nthreads_ - 1);
            for (unsigned i = 0; i + 1 < nthreads_; ++i) workers_.emplace_back([this, i] { serve(i); });
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t i = 1; i < parts; ++i) {
                const size_t off = piece * i;
                if (off >= bytes) break;
                jobs_[i - 1] = Job{static_cast<char *>(dst) + off, static_cast<const char *>(src) + off,
                                   std::min(piece, bytes - off), true};
                ++remaining_;
            }
        }
        work_.notify_all();
        std::memcpy(dst, src, std::min(piece, bytes));      // the caller takes the first piece
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return remaining_ == 0; });
        ++parallel_copies_;
    }

private:
    struct Job {
        char *dst = nullptr;
        const char *src = nullptr;
        size_t bytes = 0;
        bool pending = false;
    };

    void serve(unsigned i)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            work_.wait(lk, [&] { return stop_ || jobs_[i].pending; });
            if (stop_) return;
            const Job j = jobs_[i];
            jobs_[i].pending = false;
            lk.unlock();
            std::memcpy(j.dst, j.src, j.bytes);
            lk.lock();
            if (--remaining_ == 0) done_.notify_one();
        }
    }

    unsigned nthreads_;
    size_t parallel_from_ = kParallelFrom;
    std::mutex m_;
    std::condition_variable work_, done_;
    std::vector<std::thread> workers_;
    std::vector<Job> jobs_;
    bool stop_;
    size_t remaining_;
    size_t parallel_copies_ = 0;
};

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
    bool fits(size_t bytes) const { return bytes <= bytes_; }
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
    bool fits(size_t bytes) const { return bytes <= bytes_; }
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
    static constexpr size_t kMinBatch = 4096;          // stream samples per channel and GPU pass
    static constexpr size_t kMaxBatchCap = 1u << 20;   // ... at most (large reads), and never more than
    // this many wideband samples per pass over all channels (1 GiB of synthetic source in HBM; round 6: was 2^24, which held the
    // slow rates to passes of 2^17 (/96) and 2^18 (/48) stream samples -- 256 and 512 tiles of decim_blocks_kernel, half a
    // round and one round of the chip's workgroup slots per pass, eight and four passes per 2^20-sample read)
    static constexpr size_t kMaxSource = size_t(1) << 27;
    static constexpr size_t kGrowCap = 1u << 19;       // batches grown for a fast reader of small blocks stop here (4 MiB per copy)
    static constexpr size_t kDirectFrom = 1u << 15;    // reads at least this long are DMA-copied straight into page-locked caller memory
    // A pass whose output is at least this large lands in HBM and crosses PCIe as ONE DMA-engine copy behind the
    // kernel (57 GB/s on the boxes measured); smaller ones are stored across PCIe by the kernel itself, into the
    // chain's own hipHostMalloc'ed staging slot and nowhere else (no copy to queue, lowest latency, ~33 GB/s; the
    // mechanism -- kernel stores, an event, the host reads -- probed clean over 10^6 launches per cell,
    // tools/hostvis_probe.hip, profiles/round4c_hostvis_probe.txt).
    static constexpr size_t kSdmaFromBytes = size_t(1) << 20;

    // wire_s32: the synthetic ADC stream is S32_LE I2S words and the decimator converts them on
    // load (the reference's wire format, SoapySX.cpp:103-112); otherwise CF32 end to end.
    RxChain(int gpu, int decim, int taps_per_phase, uint64_t seed, uint32_t first_channel, int nchan, bool wire_s32)
        : gpu_(gpu), decim_(decim), ntaps_(decim * taps_per_phase), nchan_(nchan), seed_(seed),
          first_channel_(first_channel), fmt_(wire_s32 ? SXFIR_S32 : SXFIR_CF32), plan_(nullptr), next_(-1), cur_(0),
          batch_(kMinBatch)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        stream_.reset(new GpuStream());
        copy_.reset(new GpuStream());
        std::vector<float> taps((size_t)ntaps_);
        gpu_check(sxfir_design_lowpass(ntaps_, decim_, 8.0, 1.0, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_DECIMATE, taps.data(), ntaps_, decim_, nchan_, fmt_, gpu_), "sxfir_create(rx)");
        max_batch_ = kMinBatch;
        while (max_batch_ < kMaxBatchCap && 2 * max_batch_ * (size_t)decim_ * (size_t)nchan_ <= kMaxSource) max_batch_ *= 2;
        for (int k = 0; k < 2; ++k) {
            done_[k] = ran_[k] = nullptr;
            gpu_check(sxfir_event_create(&done_[k]), "sxfir_event_create");
            gpu_check(sxfir_event_create(&ran_[k]), "sxfir_event_create");
        }
        direct_done_ = nullptr;
        gpu_check(sxfir_event_create(&direct_done_), "sxfir_event_create");
    }
    ~RxChain()
    {
        sxfir_stream_sync(stream_->get());
        sxfir_stream_sync(copy_->get());
        for (int k = 0; k < 2; ++k) {
            sxfir_event_destroy(done_[k]);
            sxfir_event_destroy(ran_[k]);
        }
        sxfir_event_destroy(direct_done_);
        sxfir_destroy(plan_);
    }
    RxChain(const RxChain &) = delete;
    RxChain &operator=(const RxChain &) = delete;

    int decim() const { return decim_; }
    int ntaps() const { return ntaps_; }
    int channels() const { return nchan_; }
    int64_t direct_samples() const { return direct_samples_; }
    // what one pass of the smallest batch launches (kernel family, tiled or the generic fallback): sxfir_launch_geometry
    sxfir_geometry geometry() const
    {
        sxfir_geometry g;
        gpu_check(sxfir_launch_geometry(plan_, kMinBatch * (size_t)decim_, &g), "sxfir_launch_geometry");
        return g;
    }

    void reset()
    {
        next_ = -1;
        slot_[0].n = slot_[1].n = 0;
        grown_ = 0;
    }

    // Deliver decimated stream samples [pos, pos+n) of every channel (stream rate) to host memory as
    // interleaved CF32, dsts[c] for channel c.  Sample `m` of a stream is sum_k h[k] * source[m*decim - k];
    // the source restarts at index 0 whenever the PCMs are reset (stream position 0).
    void produce(int64_t pos, size_t n, float *const *dsts)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        // Large reads into page-locked (pinned / registered) caller memory: the pass goes from HBM into it by DMA,
        // no staging hop and no host copy.  Everything else goes through the pinned staging slots.
        dst_locked_ = n >= kDirectFrom && page_locked(dsts, n);
        // a reader of megabyte blocks into page-locked memory: the batches read ahead for it stay in HBM and are
        // DMA-copied into ITS buffer when it asks (no staging hop, no host copy)
        if (8 * n * (size_t)nchan_ >= kSdmaFromBytes) prefer_hbm_ = dst_locked_;
        const bool go_direct = dst_locked_ && (pos != next_ || n > staged_from(pos));
        if (pos != next_) {
            // a jump (overrun skip, restart): whatever was read ahead is for the wrong positions
            slot_[0].n = slot_[1].n = 0;
            prime(pos);
            if (!go_direct) {
                batch_ = pick_batch(n);
                launch(cur_, pos, batch_);
                launch(cur_ ^ 1, pos + (int64_t)batch_, batch_);
            }
        }
        size_t done = 0;
        if (go_direct) {
            // what the read-ahead already holds is handed out first: the filter has moved past it
            done = drain_slots(pos, n, dsts);
            const size_t from_slots = done;
            int64_t p = pos + (int64_t)done;
            void *st = stream_->get();
            while (done < n) {
                const size_t m = std::min(n - done, max_batch_);
                // The pass lands in HBM and a DMA-engine copy, queued behind it on the same stream, carries it into
                // the caller's memory.  (Until round 3 passes below a megabyte stored straight into the caller's
                // hipHostRegister'ed memory from the kernel.  One full-suite run then saw the last 16 bytes of one
                // tile of such a pass still unwritten when the completion event had fired -- never reproduced in
                // 1500 repeats, tools/stress_direct_rx.py -- and the copy costs nothing measurable: a kernel's own
                // stores cross PCIe at ~33 GB/s, the DMA engines write registered memory at ~36.  Kernels now store
                // only into the chain's own hipHostMalloc'ed staging, the path every small read has always taken.)
                run_to_hbm(0, p, m, st);
                for (int c = 0; c < nchan_; ++c)
                    gpu_check(sxfir_memcpy_d2h(dsts[c] + 2 * done, out_[0].at(8 * (size_t)c * m), 8 * m, st), "sxfir_memcpy_d2h");
                p += (int64_t)m;
                done += m;
            }
            gpu_check(sxfir_event_record(direct_done_, st), "sxfir_event_record");
            gpu_check(sxfir_event_sync(direct_done_), "sxfir_event_sync");
            direct_samples_ += (int64_t)(n - from_slots);
            next_ = pos + (int64_t)n;
            // keep the next batches in flight for whoever reads next
            batch_ = pick_batch(n);
            launch(cur_, next_, batch_);
            launch(cur_ ^ 1, next_ + (int64_t)batch_, batch_);
            return;
        }
        while (done < n) {
            Slot &s = slot_[cur_];
            const int64_t p = pos + (int64_t)done;
            if (s.n == 0 || p >= s.pos + (int64_t)s.n) {
                // current batch used up: the one in flight becomes current ...
                const int64_t following = s.n ? s.pos + (int64_t)s.n : p;
                cur_ ^= 1;
                if (slot_[cur_].n == 0) launch(cur_, following, batch_);
                else if (waited(cur_) && grown_ < max_batch_) {
                    // the reader caught up with the batch in flight: a caller faster than one GPU round trip per
                    // batch (no sample clock holding it back) gets batches twice as long from here on
                    grown_ = std::max(grown_, batch_) * 2;
                }
                wait(cur_);
                // ... and the batch after it goes in flight while the host hands this one out
                // (what a reader EARNS by outrunning the read-ahead stops at kGrowCap: a staged batch is one D2H copy the
                // reader of its first block waits for; only a request that is itself larger gets a larger pass, pick_batch)
                batch_ = std::min(max_batch_, std::max(pick_batch(n), std::min(grown_, kGrowCap)));
                launch(cur_ ^ 1, slot_[cur_].pos + (int64_t)slot_[cur_].n, batch_);
                continue;
            }
            const size_t off = (size_t)(p - s.pos);
            const size_t m = std::min(n - done, s.n - off);
            hand_out(cur_, off, m, dsts, done);
            done += m;
        }
        next_ = pos + (int64_t)n;
    }

private:
    struct Slot {
        int64_t pos = 0;
        size_t n = 0;         // 0 = nothing launched into this slot
        bool ready = false;   // the host may read the staged samples
        bool in_hbm = false;  // the batch is in out_[k] only: no copy to the host has been queued yet
    };

    // samples [off, off+m) of slot k -> dsts[c] + 2*done
    void hand_out(int k, size_t off, size_t m, float *const *dsts, size_t done)
    {
        Slot &s = slot_[k];
        if (s.in_hbm) {
            if (dst_locked_) {
                // The batch's kernels are long through (the host checks, it does not queue a wait: a DMA copy
                // queued behind a cross-stream wait makes hipMemcpyAsync itself block for milliseconds now and then
                // on ROCm 7.2); the copy runs on the copy stream, beside the kernels of the batches read ahead.
                void *cst = copy_->get();
                gpu_check(sxfir_event_sync(ran_[k]), "sxfir_event_sync");
                for (int c = 0; c < nchan_; ++c)
                    gpu_check(sxfir_memcpy_d2h(dsts[c] + 2 * done, out_[k].at(8 * ((size_t)c * s.n + off)), 8 * m, cst),
                              "sxfir_memcpy_d2h");
                gpu_check(sxfir_event_record(done_[k], cst), "sxfir_event_record");
                gpu_check(sxfir_event_sync(done_[k]), "sxfir_event_sync");
                direct_samples_ += (int64_t)m;
                return;
            }
            // an ordinary destination after all: the batch takes the staging hop now
            void *st = stream_->get();
            grow(stage_[k], 8 * s.n * (size_t)nchan_);
            gpu_check(sxfir_memcpy_d2h(stage_[k].floats(), out_[k].get(), 8 * s.n * (size_t)nchan_, st), "sxfir_memcpy_d2h");
            gpu_check(sxfir_event_record(done_[k], st), "sxfir_event_record");
            s.in_hbm = false;
            s.ready = false;
        }
        wait(k);
        for (int c = 0; c < nchan_; ++c)
            pool_.copy(dsts[c] + 2 * done, stage_[k].floats() + 2 * ((size_t)c * s.n + off), 8 * m);
    }

    size_t pick_batch(size_t request) const
    {
        size_t b = kMinBatch;
        while (b < 4 * request && b < max_batch_) b *= 2;
        return b;
    }

    // Are the caller's buffers page-locked (pinned / registered with sxfir_host_register), every channel's whole range?
    // Then the DMA engines can write them as they are (any alignment, any layout: one copy per channel).
    bool page_locked(float *const *dsts, size_t n) const
    {
        for (int c = 0; c < nchan_; ++c) {
            void *d = nullptr;
            if (sxfir_host_device_pointer(dsts[c], 8 * n, &d) != SXFIR_OK) return false;
        }
        return true;
    }

    // hand out whatever the two slots hold of [pos, pos+n), in stream order; returns the samples delivered.
    // Afterwards both slots are empty (the filter state is at the end of the last launched batch).
    size_t drain_slots(int64_t pos, size_t n, float *const *dsts)
    {
        size_t done = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int k = pass == 0 ? cur_ : cur_ ^ 1;
            Slot &s = slot_[k];
            const int64_t p = pos + (int64_t)done;
            if (s.n == 0) continue;
            if (p >= s.pos && p < s.pos + (int64_t)s.n && done < n) {
                const size_t off = (size_t)(p - s.pos);
                const size_t m = std::min(n - done, s.n - off);
                hand_out(k, off, m, dsts, done);
                done += m;
            }
        }
        // the filter has consumed the source up to the end of the furthest batch: the direct passes continue there
        int64_t frontier = pos + (int64_t)done;
        for (int k = 0; k < 2; ++k)
            if (slot_[k].n) frontier = std::max(frontier, slot_[k].pos + (int64_t)slot_[k].n);
        if (frontier != pos + (int64_t)done) {
            // read-ahead beyond what this call can use from staging (the request ends inside it, or the slots
            // are ahead of a short tail): re-prime at the hand-over point instead of skipping samples
            prime(pos + (int64_t)done);
        }
        slot_[0].n = slot_[1].n = 0;
        return done;
    }

    // stream samples [pos, pos+m) of all channels -> `out` (device-visible, channel stride out_stride samples)
    void run(int64_t pos, size_t m, float *out, size_t out_stride, void *st)
    {
        size_t n_out = 0;
        grow(in_, 8 * m * (size_t)decim_ * (size_t)nchan_);
        gpu_check(sxfir_synth_fill(in_.get(), m * (size_t)decim_, m * (size_t)decim_, nchan_, seed_, first_channel_,
                                   pos * decim_, fmt_, st),
                  "sxfir_synth_fill");
        gpu_check(sxfir_decimate(plan_, in_.get(), m * (size_t)decim_, m * (size_t)decim_, out, out_stride, &n_out, st),
                  "sxfir_decimate");
        if (n_out != m) throw std::runtime_error("rx chain: decimator produced an unexpected block size");
    }

    // stream samples [pos, pos+m) of all channels -> out_[k] in HBM (channel stride m); ran_[k] fires behind the pass
    void run_to_hbm(int k, int64_t pos, size_t m, void *st)
    {
        grow(out_[k], 8 * m * (size_t)nchan_);
        run(pos, m, static_cast<float *>(out_[k].get()), m, st);
        gpu_check(sxfir_event_record(ran_[k], st), "sxfir_event_record");
    }

    // stream samples [pos, pos+m) of all channels -> staging slot k, asynchronously on the chain's stream
    void launch(int k, int64_t pos, size_t m)
    {
        void *st = stream_->get();
        const size_t bytes = 8 * m * (size_t)nchan_;
        slot_[k].in_hbm = false;
        if (bytes >= kSdmaFromBytes && prefer_hbm_) {
            // large batch for a reader with page-locked buffers: it stays in HBM until that reader names its buffer
            run_to_hbm(k, pos, m, st);
            slot_[k].in_hbm = true;
        } else if (bytes >= kSdmaFromBytes) {
            // large batch: HBM, then one DMA-engine copy into the pinned staging buffer, queued behind the kernels
            // on the same stream
            grow(stage_[k], bytes);
            run_to_hbm(k, pos, m, st);
            gpu_check(sxfir_memcpy_d2h(stage_[k].floats(), out_[k].get(), bytes, st), "sxfir_memcpy_d2h");
            gpu_check(sxfir_event_record(done_[k], st), "sxfir_event_record");
        } else {
            grow(stage_[k], bytes);
            // the decimator stores straight into the pinned staging buffer (device-visible host memory): the
            // outputs are 1/decim of the traffic and cross PCIe as they are produced, no separate D2H copy
            run(pos, m, stage_[k].floats(), m, st);
            gpu_check(sxfir_event_record(done_[k], st), "sxfir_event_record");
        }
        slot_[k].pos = pos;
        slot_[k].n = m;
        slot_[k].ready = false;
    }

    // a buffer is replaced by a larger one only when nothing queued on the chain's stream can still use it
    template <class Buffer>
    void grow(Buffer &b, size_t bytes)
    {
        if (b.fits(bytes)) return;
        stream_->sync();
        copy_->sync();
        b.reserve(bytes);
    }

    // samples of [pos, ...) the two staging slots hold, contiguously from pos
    size_t staged_from(int64_t pos) const
    {
        size_t have = 0;
        int64_t p = pos;
        for (int pass = 0; pass < 2; ++pass) {
            const Slot &s = slot_[pass == 0 ? cur_ : cur_ ^ 1];
            if (s.n && p >= s.pos && p < s.pos + (int64_t)s.n) {
                have += (size_t)(s.pos + (int64_t)s.n - p);
                p = s.pos + (int64_t)s.n;
            }
        }
        return have;
    }

    // wait for THIS slot's pass; true when the host really had to wait for it (more than a few microseconds)
    bool waited(int k)
    {
        if (slot_[k].ready || slot_[k].in_hbm) return false;
        const auto t0 = std::chrono::steady_clock::now();
        wait(k);
        return std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(3);
    }

    // wait for THIS slot's pass only: the batch read ahead behind it stays in flight
    void wait(int k)
    {
        if (slot_[k].ready || slot_[k].in_hbm) return;     // (in HBM: ordered on the GPU, hand_out queues the copy)
        gpu_check(sxfir_event_sync(done_[k]), "sxfir_event_sync");
        slot_[k].ready = true;
    }

    // After a skip (overrun) or a restart the filter history is rebuilt from
    // the source: run the ntaps samples that precede `pos` through the plan.
    void prime(int64_t pos)
    {
        void *st = stream_->get();
        gpu_check(sxfir_reset(plan_, st), "sxfir_reset");
        const int64_t warm = (ntaps_ + decim_ - 1) / decim_;     // outputs whose inputs cover ntaps samples
        grow(scratch_, 8 * (size_t)warm * (size_t)nchan_);
        grow(in_, 8 * (size_t)(warm * decim_) * (size_t)nchan_);
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
    std::unique_ptr<GpuStream> copy_;   // DMA copies out of HBM into page-locked caller memory, beside the kernels on stream_
    DeviceBuffer out_[2];            // decimated block of a large pass, before its DMA copy to the host
    void *ran_[2];                   // recorded behind the pass that filled out_[k]
    CopyPool pool_;
    PinnedBuffer stage_[2];
    Slot slot_[2];
    void *done_[2];                  // recorded behind each slot's pass (its DMA copy, for a large one)
    void *direct_done_;
    int64_t next_;
    int cur_;
    size_t batch_, max_batch_;
    int64_t direct_samples_ = 0;     // samples that reached page-locked caller memory without a host copy
    size_t grown_ = 0;               // batch length a reader that outran the read-ahead has earned (0 = none)
    bool dst_locked_ = false;        // this call's destination is page-locked
    bool prefer_hbm_ = false;        // the last megabyte-sized read went to page-locked memory
};

class TxChain {
public:
    static constexpr size_t kMinSlotFrames = 1u << 15; // stream samples per channel and pinned slot: to begin with,
    static constexpr size_t kMaxSlotBytes = size_t(4) << 20;   // ... and grown for large writes up to this many bytes over all channels
    static constexpr size_t kFlushFrames = 4096;       // small writes are gathered up to this before a GPU pass
    static constexpr int kSlots = 4;
    // A pass whose input is at least this large crosses PCIe as DMA-engine copies into HBM before the kernels run
    // (57 GB/s on the boxes measured); smaller ones are read in place from the pinned slot by the interpolator,
    // which takes the keying count from the same staged tiles (nothing to queue, lowest latency, one PCIe read).
    // Measured in round 2 for the blocks in between (32 KiB ... 512 KiB; then 1.25 GS/s with two PCIe reads of the
    // slot per pass): a DMA threshold of 128 KiB halves their rate (the host then waits for the
    // input blocks; with four input blocks instead of two it is a wash: 64 Ki-sample writes 15 % faster, 32 Ki-sample
    // writes 20 % slower), non-temporal stores into the slot change nothing (the copy is not what limits them).
    static constexpr size_t kH2dFromBytes = size_t(1) << 20;
    static constexpr size_t kDirectFrom = 1u << 15;    // writes at least this long are taken straight from page-locked caller memory

    // wire_s32: the DAC-rate sink holds S32_LE I2S words with the transmitter-keying bits
    // (convert_tx_buffer, SoapySX.cpp:116-137, fused into the interpolator's store).
    TxChain(int gpu, int interp, int taps_per_phase, size_t ring_frames, int nchan, bool wire_s32)
        : gpu_(gpu), interp_(interp), ntaps_(interp * taps_per_phase), nchan_(nchan), plan_(nullptr),
          ring_len_(ring_frames * (size_t)interp), next_(0), accepted_(0), written_(0), slot_(0), pend_(0)
    {
        slot_frames_ = kMinSlotFrames;
        max_slot_frames_ = kMinSlotFrames;
        while (2 * max_slot_frames_ * 8 * (size_t)nchan_ <= kMaxSlotBytes) max_slot_frames_ *= 2;
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        stream_.reset(new GpuStream());
        copy_.reset(new GpuStream());
        std::vector<float> taps((size_t)ntaps_);
        // gain = interp: unity pass-band gain after zero stuffing
        gpu_check(sxfir_design_lowpass(ntaps_, interp_, 8.0, (double)interp_, taps.data()), "sxfir_design_lowpass");
        gpu_check(sxfir_create(&plan_, SXFIR_INTERPOLATE, taps.data(), ntaps_, interp_, nchan_,
                               wire_s32 ? SXFIR_S32 : SXFIR_CF32, gpu_),
                  "sxfir_create(tx)");
        ring_.reserve(8 * ring_len_ * (size_t)nchan_);
        stage_.reserve(8 * slot_frames_ * (size_t)nchan_ * kSlots);
        keyed_.reserve(64);
        zero_keyed();
        set_threshold2(thr2_);                             // the plan and the chain count with the same threshold bits
        for (int k = 0; k < kSlots; ++k) {
            busy_[k] = false;
            done_[k] = nullptr;
            gpu_check(sxfir_event_create(&done_[k]), "sxfir_event_create");
        }
        direct_done_ = nullptr;
        gpu_check(sxfir_event_create(&direct_done_), "sxfir_event_create");
        for (int j = 0; j < 2; ++j) {
            copied_[j] = used_[j] = nullptr;
            in_used_[j] = false;
            gpu_check(sxfir_event_create(&copied_[j]), "sxfir_event_create");
            gpu_check(sxfir_event_create(&used_[j]), "sxfir_event_create");
        }
    }
    ~TxChain()
    {
        sxfir_stream_sync(stream_->get());
        sxfir_stream_sync(copy_->get());
        for (int k = 0; k < kSlots; ++k) sxfir_event_destroy(done_[k]);
        sxfir_event_destroy(direct_done_);
        for (int j = 0; j < 2; ++j) {
            sxfir_event_destroy(copied_[j]);
            sxfir_event_destroy(used_[j]);
        }
        sxfir_destroy(plan_);
    }
    TxChain(const TxChain &) = delete;
    TxChain &operator=(const TxChain &) = delete;

    int interp() const { return interp_; }
    int channels() const { return nchan_; }
    sxfir_geometry geometry() const
    {
        sxfir_geometry g;
        gpu_check(sxfir_launch_geometry(plan_, 4096, &g), "sxfir_launch_geometry");
        return g;
    }
    int64_t written() const { return written_; }
    int64_t direct_samples() const { return direct_samples_; }
    size_t slot_frames() const { return slot_frames_; }
    void set_threshold2(float thr2)
    {
        thr2_ = thr2;
        gpu_check(sxfir_set_tx_threshold(plan_, thr2), "sxfir_set_tx_threshold");
    }

    // Transmitter keying of convert_tx_buffer (SoapySX.cpp:132-133): how many application samples of channel 0
    // reached the squared-magnitude threshold since the last reset.  Counted on the GPU as the staged blocks
    // pass (inside the interpolator, sxfir_interpolate_keyed, into a device counter), so writeStream makes no pass
    // of its own over the samples and the GPU reads them once.
    int64_t keyed_samples()
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        flush();
        unsigned long long v = 0;
        gpu_check(sxfir_memcpy_d2h(&v, keyed_.get(), sizeof(v), stream_->get()), "sxfir_memcpy_d2h");
        drain();
        return (int64_t)v;
    }

    void reset()
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        pend_ = 0;                  // what was not passed to the GPU yet is dropped with the rest of the sink
        drain();
        gpu_check(sxfir_reset(plan_, stream_->get()), "sxfir_reset");
        next_ = 0;
        accepted_ = 0;
        written_ = 0;
        zero_keyed();
        data_.clear();
        flush_frames_ = kFlushFrames;
    }

    // Stream samples [pos, pos+n) of every channel from host memory (srcs[c]).  Positions the
    // application skipped (snd_pcm_forward: timed gaps, underrun recovery) are silence.
    void consume(int64_t pos, size_t n, const float *const *srcs)
    {
        gpu_check(sxfir_set_device(gpu_), "sxfir_set_device");
        if (pos < accepted_) throw std::runtime_error("tx chain: position moved backwards");
        feed(nullptr, (size_t)(pos - accepted_));
        if (n >= kDirectFrom && device_visible(srcs, n)) direct(srcs, n);
        else feed(srcs, n);
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
        copy_->sync();
        stream_->sync();
        for (int k = 0; k < kSlots; ++k) busy_[k] = false;
    }

    // n stream samples per channel (srcs == nullptr: silence) are appended to the current pinned slot
    // (channel c at c*slot_frames_); the GPU pass over a slot is issued once kFlushFrames have gathered
    // or the slot is full, and is never waited for here (write-behind: the sink is only observable
    // through capture(), which flushes).
    void feed(const float *const *srcs, size_t n)
    {
        if (srcs && n >= 2 * slot_frames_ && slot_frames_ < max_slot_frames_) {
            // large writes: larger slots, so that a write is a few multi-threaded copies and a few GPU passes
            size_t f = slot_frames_;
            while (f < max_slot_frames_ && 2 * f <= n) f *= 2;
            resize_slots(f);
        }
        size_t done = 0;
        while (done < n) {
            if (pend_ == 0 && busy_[slot_]) {
                // the slot's last H2D may still be reading it: wait for THAT pass only, not for the newer ones
                const auto t0 = std::chrono::steady_clock::now();
                gpu_check(sxfir_event_sync(done_[slot_]), "sxfir_event_sync");
                busy_[slot_] = false;
                // a writer that comes round to a slot still in use is faster than one GPU pass per kFlushFrames
                // samples (no sample clock holding it back): gather twice as much per pass from here on
                if (flush_frames_ < slot_frames_ && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(3))
                    flush_frames_ *= 2;
            }
            const size_t m = std::min(slot_frames_ - pend_, n - done);
            float *host = stage_.floats() + 2 * slot_frames_ * (size_t)nchan_ * (size_t)slot_;
            for (int c = 0; c < nchan_; ++c) {
                float *dst = host + 2 * ((size_t)c * slot_frames_ + pend_);
                if (srcs) pool_.copy(dst, srcs[c] + 2 * done, 8 * m);
                else std::memset(dst, 0, 8 * m);
            }
            if (srcs) {
                // application data (not silence): these samples take part in the keying count
                if (!data_.empty() && data_.back().first + data_.back().second == pend_) data_.back().second += m;
                else data_.emplace_back(pend_, m);
            }
            pend_ += m;
            accepted_ += (int64_t)m;
            done += m;
            if (pend_ >= flush_frames_) flush();
        }
    }

    // pass the gathered samples of the current slot through the interpolator into the sink ring, asynchronously
    void flush()
    {
        if (pend_ == 0) return;
        void *st = stream_->get();
        const size_t slot_off = slot_frames_ * (size_t)nchan_ * (size_t)slot_;
        const char *host = reinterpret_cast<const char *>(stage_.floats() + 2 * slot_off);
        if (8 * pend_ * (size_t)nchan_ >= kH2dFromBytes) {
            // large pass: DMA-engine copies into HBM on the copy stream, beside the kernels of the pass before; the
            // slot is free again as soon as they are through
            void *cst = begin_copy();
            for (int c = 0; c < nchan_; ++c)
                gpu_check(sxfir_memcpy_h2d(in_dev_[cur_in_].at(8 * (size_t)c * max_slot_frames_),
                                           host + 8 * (size_t)c * slot_frames_, 8 * pend_, cst),
                          "sxfir_memcpy_h2d");
            gpu_check(sxfir_event_record(done_[slot_], cst), "sxfir_event_record");
            pass_copied(pend_, st);
        } else {
            // the interpolator reads its input straight from the pinned slot (device-visible host memory): the
            // input is 1/interp of the traffic and crosses PCIe as the kernel fetches it, no separate H2D copy
            pass(host, slot_frames_, pend_, st);
            gpu_check(sxfir_event_record(done_[slot_], st), "sxfir_event_record");
        }
        busy_[slot_] = true;
        slot_ = (slot_ + 1) % kSlots;
        pend_ = 0;
    }

    // `frames` stream samples per channel at `base` (device-visible, channel stride `stride` samples) -> the
    // interpolator -> sink ring at ring position next_*interp (in two passes where the ring wraps); the ranges of
    // data_ (application data, relative to base) take part in the keying count
    void pass(const char *base, size_t stride, size_t frames, void *st)
    {
        // One range of application data (the usual case: a slot is all data, or data behind a timed gap): the
        // interpolator takes the keying count itself from the tile it has staged in LDS, so a pass read in place from
        // the pinned slot crosses PCIe once (round 3: a second kernel read the slot again, 1.2-1.8 GS/s for 4096 ...
        // 65536-sample writes).  Several ranges in one slot (data, silence, data): the separate count kernel.
        const bool fused = data_.size() == 1;
        const size_t k_lo = fused ? data_[0].first : 0, k_hi = fused ? data_[0].first + data_[0].second : 0;
        size_t done = 0;
        while (done < frames) {
            const size_t off = (size_t)((next_ * interp_) % (int64_t)ring_len_);
            const size_t m = std::min(frames - done, (ring_len_ - off) / (size_t)interp_);
            size_t n_out = 0;
            // this call's share of the keying range, relative to its first sample
            const size_t lo = std::min(std::max(k_lo, done), done + m) - done, hi = std::min(std::max(k_hi, done), done + m) - done;
            if (fused && hi > lo)
                gpu_check(sxfir_interpolate_keyed(plan_, base + 8 * done, m, stride, ring_.at(8 * off), ring_len_, &n_out, lo,
                                                  hi - lo, keyed_counter(), st),
                          "sxfir_interpolate_keyed");
            else
                gpu_check(sxfir_interpolate(plan_, base + 8 * done, m, stride, ring_.at(8 * off), ring_len_, &n_out, st),
                          "sxfir_interpolate");
            next_ += (int64_t)m;
            done += m;
        }
        if (!fused)
            for (const auto &r : data_)
                gpu_check(sxfir_count_keyed(reinterpret_cast<const float *>(base + 8 * r.first), r.second, thr2_, keyed_counter(), st),
                          "sxfir_count_keyed");
        data_.clear();
    }

    // Are the caller's blocks page-locked (pinned / registered with sxfir_host_register)?  Then the DMA engines
    // can take them as they are.
    bool device_visible(const float *const *srcs, size_t n) const
    {
        for (int c = 0; c < nchan_; ++c) {
            void *d = nullptr;
            if (sxfir_host_device_pointer(srcs[c], 8 * n, &d) != SXFIR_OK) return false;
        }
        return true;
    }

    // A large write from page-locked caller memory: no staging copy.  The blocks go to HBM by DMA and through the
    // kernels from there; the call returns when the last copy has read the caller's memory (the kernels run on).
    void direct(const float *const *srcs, size_t n)
    {
        flush();                                           // what was gathered before comes first
        void *st = stream_->get();
        size_t done = 0;
        while (done < n) {
            const size_t m = std::min(n - done, max_slot_frames_);
            void *cst = begin_copy();
            for (int c = 0; c < nchan_; ++c)
                gpu_check(sxfir_memcpy_h2d(in_dev_[cur_in_].at(8 * (size_t)c * max_slot_frames_), srcs[c] + 2 * done, 8 * m, cst),
                          "sxfir_memcpy_h2d");
            if (done + m == n) gpu_check(sxfir_event_record(direct_done_, cst), "sxfir_event_record");
            data_.emplace_back(0, m);
            pass_copied(m, st);
            done += m;
        }
        gpu_check(sxfir_event_sync(direct_done_), "sxfir_event_sync");
        accepted_ += (int64_t)n;
        direct_samples_ += (int64_t)n;
    }

    // The next input block in HBM (two alternate): it may be filled once the kernels that read its last contents
    // are through.  Returns the copy stream.
    void *begin_copy()
    {
        grow(in_dev_[cur_in_], 8 * max_slot_frames_ * (size_t)nchan_);
        // (the HOST waits here, not the copy stream: the application thread is then never more than two passes
        // ahead of the GPU.  With the wait queued on the copy stream instead, a caller that runs ahead makes
        // hipMemcpyAsync itself block for 7-8 ms now and then on ROCm 7.2 -- measured, tools/devpath_probe.py.)
        if (in_used_[cur_in_]) gpu_check(sxfir_event_sync(used_[cur_in_]), "sxfir_event_sync");
        return copy_->get();
    }

    // ... and the kernels over it, once the copies queued since begin_copy() are through
    void pass_copied(size_t frames, void *st)
    {
        gpu_check(sxfir_event_record(copied_[cur_in_], copy_->get()), "sxfir_event_record");
        gpu_check(sxfir_stream_wait_event(st, copied_[cur_in_]), "sxfir_stream_wait_event");
        pass(in_dev_[cur_in_].at(0), max_slot_frames_, frames, st);
        gpu_check(sxfir_event_record(used_[cur_in_], st), "sxfir_event_record");
        in_used_[cur_in_] = true;
        cur_in_ ^= 1;
    }

    void resize_slots(size_t frames)
    {
        flush();
        drain();
        slot_frames_ = frames;
        stage_.reserve(8 * slot_frames_ * (size_t)nchan_ * kSlots);
        slot_ = 0;
    }

    // a buffer is replaced by a larger one only when nothing queued on the chain's stream can still use it
    template <class Buffer>
    void grow(Buffer &b, size_t bytes)
    {
        if (b.fits(bytes)) return;
        copy_->sync();
        stream_->sync();
        b.reserve(bytes);
    }

    unsigned long long *keyed_counter() const { return static_cast<unsigned long long *>(keyed_.get()); }
    void zero_keyed()
    {
        static const unsigned long long zero = 0;
        gpu_check(sxfir_memcpy_h2d(keyed_.get(), &zero, sizeof(zero), stream_->get()), "sxfir_memcpy_h2d");
        stream_->sync();
    }

    int gpu_, interp_, ntaps_, nchan_;
    sxfir_plan *plan_;
    size_t ring_len_;
    std::unique_ptr<GpuStream> stream_;
    DeviceBuffer ring_;
    PinnedBuffer stage_;
    std::unique_ptr<GpuStream> copy_;                     // DMA copies of large passes, beside the kernels on stream_
    DeviceBuffer in_dev_[2];                              // input blocks of large passes (DMA-copied from the slot or the caller)
    void *copied_[2], *used_[2];                          // in_dev_[j] filled (copy stream) / read by its kernels (stream_)
    bool in_used_[2];
    int cur_in_ = 0;
    CopyPool pool_;
    size_t slot_frames_, max_slot_frames_;
    size_t flush_frames_ = kFlushFrames;                  // samples gathered before a GPU pass; grows for writers that outrun the passes
    void *direct_done_;
    int64_t direct_samples_ = 0;                          // samples taken straight from page-locked caller memory
    DeviceBuffer keyed_;                                  // the keying counter (device memory, read back on demand)
    float thr2_ = 1.0e-6f;
    std::vector<std::pair<size_t, size_t>> data_;         // (offset, length) of application data in the current slot
    bool busy_[kSlots];
    void *done_[kSlots];  // recorded behind each slot's GPU pass
    int64_t next_;        // stream samples passed to the GPU so far (written + silence)
    int64_t accepted_;    // next_ + what is gathered in the current slot
    int64_t written_;     // stream samples that carried application data
    int slot_;
    size_t pend_;         // samples gathered in the current slot and not yet passed to the GPU
};

}  // namespace sx
