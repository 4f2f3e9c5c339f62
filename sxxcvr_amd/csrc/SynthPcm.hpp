// SynthPcm: the synthetic sample clock + ring model that replaces the ALSA/I2S
// PCM of the reference (class AlsaPcm, SoapySX.cpp:369-518, and the snd_pcm_*
// calls made from readStream/writeStream, SoapySX.cpp:897-1104).
//
// It keeps the same counters with the same meaning (position, period size,
// buffer size, stream mode, setup_done, activated, per-stream mutex) and offers
// the small set of PCM operations the stream code needs, with ALSA's
// semantics for them:
//   avail_delay  capture:  avail = hw - appl, delay = avail
//                playback: delay = appl - hw, avail = buffer - delay
//   NORMAL mode  never stops on xrun (stop_threshold = boundary) and plays
//                silence for what was not written (SoapySX.cpp:492-496)
//   LINK mode    stops both linked PCMs on xrun (stop_threshold = buffer) and
//                starts them on the first TX write (SoapySX.cpp:36-43,497-501)
// Time comes from a SampleClock: "virtual" (deterministic; advanced explicitly
// or by blocking calls) or "wall" (free-running steady clock at the sample
// rate).  Sample data comes from / goes to the GPU chains (GpuChains.hpp).
#pragma once

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <mutex>
#include <thread>

namespace sx {

class SampleClock {
public:
    enum Mode { VIRTUAL, WALL };

    SampleClock(Mode mode, double rate) : mode_(mode), rate_(rate), base_(0), t0_(std::chrono::steady_clock::now()) {}

    Mode mode() const { return mode_; }

    // Samples elapsed since construction (monotonic).
    int64_t now() const
    {
        std::lock_guard<std::mutex> lock(m_);
        return now_locked();
    }

    // Virtual clock only: let `n` sample periods pass.
    void advance(int64_t n)
    {
        std::lock_guard<std::mutex> lock(m_);
        if (mode_ == VIRTUAL && n > 0) base_ += n;
    }

    // Block until the clock reads at least `target` samples.
    void wait_until(int64_t target)
    {
        if (mode_ == VIRTUAL) {
            std::lock_guard<std::mutex> lock(m_);
            if (base_ < target) base_ = target;
            return;
        }
        for (;;) {
            const int64_t n = now();
            if (n >= target) return;
            const double secs = (double)(target - n) / rate_;
            std::this_thread::sleep_for(std::chrono::duration<double>(secs > 0.0005 ? secs * 0.9 : 0.00005));
        }
    }

    // RX and TX may be driven from different threads (per-stream mutexes, SoapySX.cpp:373), but linked
    // PCMs start and stop each other: those state changes are serialised here.
    std::recursive_mutex &link_mutex() { return link_; }

    // The sample rate changed: keep the count, change the slope.
    void set_rate(double rate)
    {
        std::lock_guard<std::mutex> lock(m_);
        base_ = now_locked();
        t0_ = std::chrono::steady_clock::now();
        rate_ = rate;
    }

private:
    int64_t now_locked() const
    {
        if (mode_ == VIRTUAL) return base_;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
        return base_ + (int64_t)(dt * rate_);
    }

    Mode mode_;
    double rate_;
    int64_t base_;
    std::chrono::steady_clock::time_point t0_;
    mutable std::mutex m_;
    std::recursive_mutex link_;
};

enum stream_mode { STREAM_MODE_NORMAL, STREAM_MODE_LINK };

class SynthPcm {
public:
    enum State { PREPARED, RUNNING, XRUN };
    enum Dir { CAPTURE, PLAYBACK };

    const char *name;
    Dir dir;
    mutable std::mutex mutex;
    enum stream_mode stream_mode;
    bool setup_done;
    bool activated;
    // samples handed to / taken from the application, incl. skipped ones.  Written by the stream's own calls under the
    // stream mutex (as in the reference, :373); atomic because the build's RX_POSITION / TX_POSITION settings read it from
    // any thread without queueing behind a blocking read or write.
    std::atomic<int64_t> position;
    uint64_t hwp_period_size;
    uint64_t hwp_buffer_size;

    SynthPcm(const char *name_, Dir dir_, SampleClock *clock)
        : name(name_), dir(dir_), stream_mode(STREAM_MODE_NORMAL), setup_done(false), activated(false), position(0),
          hwp_period_size(0), hwp_buffer_size(0), clock_(clock), peer_(nullptr), state_(PREPARED), start_clock_(0),
          appl_(0), hw_frozen_(0)
    {
    }

    bool is_tx() const { return dir == PLAYBACK; }
    State state() const
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        return state_;
    }
    int64_t appl() const { return appl_; }

    void link(SynthPcm *peer)
    {
        peer_ = peer;
        peer->peer_ = this;
    }
    bool linked() const { return peer_ != nullptr; }

    // Geometry of AlsaPcm::configure (SoapySX.cpp:451-466): period defaults to
    // 256, is capped at 65536, and the ring is the largest multiple of the
    // period that fits in 65536 frames.
    void configure(uint64_t period)
    {
        const uint64_t max_buffer_size = 65536;
        hwp_period_size = period > 0 ? period : 256;
        if (hwp_period_size > max_buffer_size) hwp_period_size = max_buffer_size;
        hwp_buffer_size = max_buffer_size / hwp_period_size * hwp_period_size;
        reset();
    }

    // drop + prepare + reset (SoapySX.cpp:419-432)
    int reset()
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        state_ = PREPARED;
        appl_ = 0;
        hw_frozen_ = 0;
        position = 0;
        return 0;
    }

    // snd_pcm_start; linked PCMs start on the same clock tick
    int start()
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        const int64_t t = clock_->now();
        start_one(t);
        if (peer_) peer_->start_one(t);
        return 0;
    }

    int64_t hw() const
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        return state_ == RUNNING ? clock_->now() - start_clock_ : hw_frozen_;
    }

    // Every operation below reads (and some change) state_ / start_clock_ / hw_frozen_, which a LINKED peer driven from
    // another thread starts and stops too (SoapySX.cpp:36-43: snd_pcm_link): they hold the link mutex while they look,
    // and never while they wait for the clock.

    // snd_pcm_avail_delay.  Returns 0 or -EPIPE (stopped by an xrun, LINK mode).
    int avail_delay(int64_t *avail, int64_t *delay)
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        check_xrun();
        if (state_ == XRUN) return -EPIPE;
        const int64_t h = hw();
        if (dir == CAPTURE) {
            *avail = h - appl_;
            *delay = *avail;
        } else {
            *delay = appl_ - h;
            *avail = (int64_t)hwp_buffer_size - *delay;
        }
        return 0;
    }

    // snd_pcm_forwardable / snd_pcm_forward: move the application pointer
    // without transferring data.
    int64_t forwardable()
    {
        int64_t avail = 0, delay = 0;
        const int rc = avail_delay(&avail, &delay);
        if (rc < 0) return rc;
        return avail > 0 ? avail : 0;
    }

    int64_t forward(int64_t frames)
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        const int64_t can = forwardable();
        if (can < 0) return can;
        if (frames > can) frames = can;
        if (frames < 0) frames = 0;
        appl_ += frames;
        return frames;
    }

    // snd_pcm_wait: block until at least one frame can be transferred.
    void wait_for_space_or_data()
    {
        int64_t target;
        {
            std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
            if (state_ != RUNNING) return;
            target = dir == CAPTURE ? start_clock_ + appl_ + 1 : start_clock_ + appl_ - (int64_t)hwp_buffer_size + 1;
        }
        clock_->wait_until(target);
    }

    // Blocking transfer bookkeeping of snd_pcm_readi: waits (lets the clock
    // run) until `frames` are available, returns the stream index of the first
    // frame in *first and advances the application pointer.  A capture PCM
    // that was only prepared starts on its first read (start_threshold = 1).
    int64_t begin_read(int64_t frames, int64_t *first)
    {
        int64_t target;
        {
            std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
            if (state_ == PREPARED) start();
            check_xrun();
            if (state_ == XRUN) return -EPIPE;
            target = start_clock_ + appl_ + frames;
        }
        clock_->wait_until(target);
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        check_xrun();
        if (state_ == XRUN) return -EPIPE;
        *first = appl_;
        appl_ += frames;
        return frames;
    }

    // Bookkeeping of snd_pcm_writei: waits until the ring has room for all
    // `frames`, returns the index of the first frame.  In LINK mode the first
    // write starts the (linked) PCMs (start_threshold = 1).
    int64_t begin_write(int64_t frames, int64_t *first)
    {
        for (;;) {
            int64_t target = 0;
            bool running;
            {
                std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
                check_xrun();
                if (state_ == XRUN) return -EPIPE;
                running = state_ == RUNNING;
                if (running) target = start_clock_ + appl_ + frames - (int64_t)hwp_buffer_size;
            }
            if (running) clock_->wait_until(target);
            std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
            if (running) {
                // (no xrun evaluation of its own here: a call that waited for room has by construction not run dry; but a
                // linked capture PCM that overflowed meanwhile has stopped this one too)
                if (state_ == XRUN) return -EPIPE;
            } else if (state_ == RUNNING) {
                // the linked capture thread's first read started both PCMs between the two looks: this call neither waited
                // for room (it saw PREPARED) nor may it take the not-running clamp below (the ring drains now) -- look again
                // and wait as a running PCM does, where ALSA would block
                continue;
            } else {
                const int64_t room = (int64_t)hwp_buffer_size - appl_;
                if (frames > room) frames = room > 0 ? room : 0;   // not running: cannot drain
            }
            *first = appl_;
            appl_ += frames;
            if (state_ == PREPARED && frames > 0 && stream_mode == STREAM_MODE_LINK) start();
            return frames;
        }
    }

private:
    void start_one(int64_t t)
    {
        if (state_ != PREPARED) return;
        state_ = RUNNING;
        start_clock_ = t - hw_frozen_;
    }

    void stop_one()
    {
        if (state_ != RUNNING) return;
        hw_frozen_ = clock_->now() - start_clock_;
        state_ = XRUN;
    }

    // LINK mode only: the ring overflowed (capture) or ran dry (playback).
    void check_xrun()
    {
        std::lock_guard<std::recursive_mutex> lk(clock_->link_mutex());
        if (state_ != RUNNING || stream_mode != STREAM_MODE_LINK) return;
        const int64_t h = clock_->now() - start_clock_;
        const bool xrun = (dir == CAPTURE) ? (h - appl_ >= (int64_t)hwp_buffer_size) : (appl_ - h <= 0);
        if (!xrun) return;
        stop_one();
        if (peer_) peer_->stop_one();
    }

    SampleClock *clock_;
    SynthPcm *peer_;
    State state_;
    int64_t start_clock_;
    int64_t appl_;
    int64_t hw_frozen_;
};

}  // namespace sx
