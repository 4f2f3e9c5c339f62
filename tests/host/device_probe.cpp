// CPU-only probe of the Device module as applications drive it -- several threads at once -- through the flat C view
// (include/sx_device.h) over the test-only ASYNCHRONOUS fake backend (fake_sxfir.cpp): SoapySXHip.cpp +
// sx_device_capi.cpp + GpuChains.hpp + SynthPcm.hpp compiled as they ship, no GPU.  tests/test_host_logic.py builds it
// plain, with -fsanitize=address,undefined and with -fsanitize=thread, and runs all three.
//
// The reference's threading contract (SoapySX/SoapySX.cpp): any thread may call; RX and TX run on different threads
// (example/plot_rxtx_response.py:65-77) under per-stream mutexes held for the whole call (:373, :878, :979);
// getHardwareTime contends with TX only (:1110-1125); lifecycle calls take both (:750, :806, :835).
//
// Scenarios (the patterns of tests/test_gpu_device.py: test_rx_and_tx_threads, ..._with_megabyte_blocks,
// test_random_call_sequences_keep_the_stream_intact, test_linked_streams):
//   threads     RX thread + TX thread + a third thread polling getHardwareTime / getSampleRate / settings / registers,
//               wall clock, 4096-sample blocks; RX data and timestamps against the oracle
//   megabyte    the same with 2^18-sample blocks, page-locked and ordinary buffers alternating (DMA paths, copy pools,
//               second streams of both chains at once)
//   linked      link=1 streams from two threads: started by the first TX write, stopped together by the underrun,
//               restarted by deactivate / activate while the other thread keeps calling
//   calls       N small calls on the virtual clock (timed writes, non-blocking reads, overrun skips, resets) with the
//               poller running: every RX block is the oracle's at the position its timestamp names
#include <sx_device.h>
#include <sxfir.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
#include "sx_oracle.h"
}

static std::atomic<int> bad{0};
static void expect(bool ok, const char *what)
{
    if (!ok) {
        std::printf("FAIL %s (%s)\n", what, sx_device_last_error());
        ++bad;
    }
}

static const uint64_t SEED = 0x51255;

// decimated reference stream of channel 0 from position 0 (what every RX block must be a slice of)
struct RxRef {
    int D;
    std::vector<float> taps, y;
    RxRef(int D_, size_t n_out) : D(D_), taps(32 * D_), y(2 * n_out)
    {
        const int NT = 32 * D;
        std::vector<float> x(2 * n_out * D);
        sxo_design_lowpass(NT, D, 8.0, 1.0, taps.data());
        sxo_synth_iq(SEED, 0, 0, n_out * D, x.data());
        sxo_decim_f32(taps.data(), NT, D, 2, 4, x.data(), n_out * D, 0, n_out, y.data());
    }
    // is `got` the stream's samples [pos, pos + n)?  Inside the precomputed window a memcmp; beyond it (a slow host on
    // the wall clock is skipped far ahead by the overrun rule) the block is filtered on the spot from the source
    // samples that feed it: the data check never depends on how fast this machine is
    bool holds(long long pos, size_t n, const float *got) const
    {
        if (pos < 0) return false;
        if (2 * ((size_t)pos + n) <= y.size()) return std::memcmp(got, y.data() + 2 * pos, 8 * n) == 0;
        const int NT = 32 * D;
        const long long first_in = pos * D - NT;                 // NT samples of history in front (a multiple of D)
        if (first_in < 0) return false;
        std::vector<float> x(2 * ((size_t)NT + n * D)), want(2 * n);
        sxo_synth_iq(SEED, 0, first_in, (size_t)NT + n * D, x.data());
        sxo_decim_f32(taps.data(), NT, D, 2, 4, x.data(), (size_t)NT + n * D, NT / D, n, want.data());
        return std::memcmp(got, want.data(), 8 * n) == 0;
    }
};

struct Dev {
    sx_device *d = nullptr;
    sx_stream *rx = nullptr, *tx = nullptr;
    double rate;
    Dev(const char *args, double rate_, const char *rx_args, const char *tx_args) : rate(rate_)
    {
        d = sx_device_make(args);
        expect(d != nullptr, "make");
        if (!d) std::exit(2);
        expect(sx_device_set_sample_rate(d, SX_SOAPY_SDR_RX, 0, rate) == 0, "set_sample_rate");
        const size_t ch0 = 0;
        rx = sx_device_setup_stream(d, SX_SOAPY_SDR_RX, "CF32", &ch0, 1, rx_args);
        tx = sx_device_setup_stream(d, SX_SOAPY_SDR_TX, "CF32", &ch0, 1, tx_args);
        expect(rx && tx, "setup_stream");
    }
    ~Dev()
    {
        sx_device_close_stream(d, rx);
        sx_device_close_stream(d, tx);
        sx_device_unmake(d);
    }
    long long setting(const char *key)
    {
        char out[64] = "";
        sx_device_read_setting(d, key, out, sizeof(out));
        return std::atoll(out);
    }
};

// the third thread of every scenario: calls that take the TX mutex, the register mutex, or none
static void poller(Dev *dev, std::atomic<bool> *stop, std::atomic<long> *polls)
{
    long long last = -1;
    unsigned k = 0;
    while (!stop->load()) {
        long long t = 0;
        if (sx_device_get_hardware_time(dev->d, "", &t) == 0) {
            (void)last;
            last = t;
        }
        (void)sx_device_get_sample_rate(dev->d, SX_SOAPY_SDR_RX, 0);
        (void)dev->setting("RX_POSITION");
        (void)dev->setting("TX_POSITION");
        if ((k & 7) == 0) {
            sx_device_set_gain(dev->d, SX_SOAPY_SDR_RX, 0, (double)(k % 40));
            (void)sx_device_get_gain(dev->d, SX_SOAPY_SDR_RX, 0);
            sx_device_set_frequency(dev->d, SX_SOAPY_SDR_TX, 0, 433.0e6 + 1000.0 * (k % 100));
            unsigned v[4];
            sx_device_read_registers(dev->d, "", 0, v, 4);
        }
        (void)sx_device_get_stream_mtu(dev->d, dev->rx);
        ++k;
        ++*polls;
        // (not a spin: std::mutex is not fair, a poller that re-takes the TX mutex at once starves the stream threads)
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// RX thread + TX thread + poller on the wall clock; blocks of `blk`; every other block through page-locked memory
// when `pinned`
static void scenario_threads(const char *name, size_t blk, int nblk, bool pinned, bool allow_gaps)
{
    const double rate = 600000.0;
    std::string period = "period=" + std::to_string(blk > 65536 ? (size_t)65536 : blk);
    Dev dev("driver=sx,clock=wall", rate, period.c_str(), (period + ",threshold=0.5").c_str());
    RxRef ref(4, (size_t)nblk * blk + 4096);
    expect(sx_device_activate_stream(dev.d, dev.rx, 0, 0, 0) == 0, "activate rx");
    expect(sx_device_activate_stream(dev.d, dev.tx, 0, 0, 0) == 0, "activate tx");
    std::atomic<bool> stop{false};
    std::atomic<long> polls{0};
    std::thread third(poller, &dev, &stop, &polls);

    std::vector<float> sent(2 * blk);
    sxo_synth_iq(77, 3, 0, blk, sent.data());
    long long keyed_per_block = 0;
    for (size_t i = 0; i < blk; ++i) keyed_per_block += (sent[2 * i] * sent[2 * i] + sent[2 * i + 1] * sent[2 * i + 1] >= 0.25f) ? 1 : 0;
    void *pin_tx = nullptr, *pin_rx = nullptr;
    if (pinned) {
        sxfir_host_alloc(&pin_tx, 8 * blk);
        sxfir_host_alloc(&pin_rx, 8 * blk);
        std::memcpy(pin_tx, sent.data(), 8 * blk);
    }
    std::thread txt([&] {
        for (int i = 0; i < nblk; ++i) {
            const void *b[1] = {(pinned && (i & 1)) ? pin_tx : (const void *)sent.data()};
            int flags = 0;
            const int r = sx_device_write_stream(dev.d, dev.tx, b, blk, &flags, 0, 1000000);
            expect(r == (int)blk, "writeStream returns the block");
        }
    });
    std::vector<float> plain(2 * blk);
    long long want_pos = 0;
    int bad_blocks = 0, gaps = 0;
    for (int i = 0; i < nblk; ++i) {
        float *buf = (pinned && (i & 1)) ? static_cast<float *>(pin_rx) : plain.data();
        void *b[1] = {buf};
        int flags = 0;
        long long t_ns = -1;
        const int r = sx_device_read_stream(dev.d, dev.rx, b, blk, &flags, &t_ns, 1000000);
        expect(r == (int)blk && (flags & SX_SOAPY_SDR_HAS_TIME), "readStream returns the block with a timestamp");
        const long long pos = sx_time_ns_to_ticks(t_ns, rate);
        // (on the wall clock a host that takes longer than the 65536-sample ring per block is skipped ahead by the
        // overrun rule, :910-927 -- the oracle-backed fake is that slow for megabyte blocks; the data must still be the
        // stream's at the position the timestamp names)
        if (pos != want_pos) ++gaps;
        if (pos < want_pos || !ref.holds(pos, blk, buf)) ++bad_blocks;
        want_pos = pos + (long long)blk;
    }
    txt.join();
    stop = true;
    third.join();
    expect(bad_blocks == 0, "every RX block is the oracle's at the position its timestamp names");
    expect(allow_gaps || gaps == 0, "contiguous from position 0");
    expect(dev.setting("TX_WRITTEN") == (long long)nblk * (long long)blk, "TX_WRITTEN");
    expect(dev.setting("TX_PTT_SAMPLES") == (long long)nblk * keyed_per_block, "TX_PTT_SAMPLES");
    std::printf("%s blocks %d bad %d overrun_skips %d polls %ld\n", name, nblk, bad_blocks, gaps, polls.load());
    expect(sx_device_deactivate_stream(dev.d, dev.rx, 0, 0) == 0, "deactivate rx");
    expect(sx_device_deactivate_stream(dev.d, dev.tx, 0, 0) == 0, "deactivate tx");
    if (pinned) {
        sxfir_host_free(pin_tx);
        sxfir_host_free(pin_rx);
    }
}

// link=1: the PCMs start on the first TX write and stop together when the playback ring runs dry
// (SoapySX.cpp:36-43, :497-501; test/test_linked_streams.py); RX and TX threads and the poller run through the
// start, the xrun and a restart.
static void scenario_linked(int rounds)
{
    const double rate = 600000.0;
    const size_t blk = 1024;
    Dev dev("driver=sx,clock=wall", rate, "period=1024,link=1", "period=1024,link=1,threshold=0");
    RxRef ref(4, 400000);
    std::atomic<bool> stop{false};
    std::atomic<long> polls{0};
    std::thread third(poller, &dev, &stop, &polls);
    int xruns_seen = 0, bad_blocks = 0;
    for (int round = 0; round < rounds; ++round) {
        expect(sx_device_activate_stream(dev.d, dev.rx, 0, 0, 0) == 0, "activate rx (linked)");
        expect(sx_device_activate_stream(dev.d, dev.tx, 0, 0, 0) == 0, "activate tx (linked)");
        std::atomic<bool> tx_done{false};
        std::thread txt([&] {
            std::vector<float> z(2 * blk, 0.25f);
            const void *b[1] = {z.data()};
            for (int i = 0; i < 24; ++i) {
                int flags = 0;
                const int r = sx_device_write_stream(dev.d, dev.tx, b, blk, &flags, 0, 100000);
                if (r < 0) break;
            }
            tx_done = true;            // ... and stops writing: the ring (24 blocks = 41 ms) runs dry, both PCMs stop
        });
        std::vector<float> buf(2 * blk);
        void *b[1] = {buf.data()};
        int got_blocks = 0;
        bool stopped = false;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(20);
        while (!stopped && std::chrono::steady_clock::now() < deadline) {
            int flags = 0;
            long long t_ns = 0;
            const int r = sx_device_read_stream(dev.d, dev.rx, b, blk, &flags, &t_ns, 0);     // non-blocking
            if (r == SX_SOAPY_SDR_OVERFLOW || r == SX_SOAPY_SDR_STREAM_ERROR) {
                stopped = true;
            } else if (r > 0) {
                const long long pos = sx_time_ns_to_ticks(t_ns, rate);
                if (!ref.holds(pos, (size_t)r, buf.data())) ++bad_blocks;
                ++got_blocks;
            } else {
                std::this_thread::sleep_for(std::chrono::microseconds(500));
            }
        }
        txt.join();
        xruns_seen += stopped ? 1 : 0;
        expect(got_blocks > 0, "linked RX delivered blocks after the TX write started the PCMs");
        // both inactive -> reset: the next round starts at position 0 again (:850-854)
        expect(sx_device_deactivate_stream(dev.d, dev.rx, 0, 0) == 0, "deactivate rx (linked)");
        expect(sx_device_deactivate_stream(dev.d, dev.tx, 0, 0) == 0, "deactivate tx (linked)");
        expect(dev.setting("RX_POSITION") == 0 && dev.setting("TX_POSITION") == 0, "positions rewound");
    }
    stop = true;
    third.join();
    expect(bad_blocks == 0, "linked RX blocks are the oracle's");
    expect(xruns_seen == rounds, "the underrun stopped both PCMs in every round");
    std::printf("linked rounds %d xruns %d bad %d polls %ld\n", rounds, xruns_seen, bad_blocks, polls.load());
}

static uint64_t rnd_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd()
{
    rnd_state ^= rnd_state << 13;
    rnd_state ^= rnd_state >> 7;
    rnd_state ^= rnd_state << 17;
    return rnd_state;
}

// Many small calls on the virtual clock, one application thread + the poller: blocking and non-blocking reads, reads
// after the clock ran ahead (overrun skip: a jump of the chain), timed writes ahead and in the past, untimed writes after
// an underrun, now and then both streams deactivated and activated again (reset).  Every RX block must be the oracle's
// at the position its timestamp names; every accepted TX sample is accounted for.
static void scenario_calls(long ncalls)
{
    const double rate = 600000.0;
    Dev dev("driver=sx,clock=virtual", rate, "period=256", "period=256,threshold=0");
    const size_t ref_len = 3000000;
    RxRef ref(4, ref_len);
    std::atomic<bool> stop{false};
    std::atomic<long> polls{0};
    std::thread third(poller, &dev, &stop, &polls);
    expect(sx_device_activate_stream(dev.d, dev.rx, 0, 0, 0) == 0, "activate rx");
    expect(sx_device_activate_stream(dev.d, dev.tx, 0, 0, 0) == 0, "activate tx");
    std::vector<float> buf(2 * 8192), out(2 * 8192, 0.125f);
    void *pin = nullptr;
    sxfir_host_alloc(&pin, 8 * 70000);
    long bad_blocks = 0, reads = 0, writes = 0, skips = 0, resets = 0;
    long long tx_accepted = 0;
    for (long call = 0; call < ncalls; ++call) {
        unsigned op = (unsigned)(rnd() % 100);
        // (a jump or a restart throws the read-ahead away, up to two batches of 2^20 samples the oracle computed for
        // nothing: one call in fifty, so that 10^5 calls stay minutes)
        const unsigned rare = (unsigned)(rnd() % 1000);
        if (op >= 90 && rare >= 200) op = 10 + rare % 80;
        const long long rx_pos = dev.setting("RX_POSITION");
        if (rx_pos > (long long)ref_len - 300000 || rare < 2) {
            // restart: both inactive -> positions and chains reset
            sx_device_deactivate_stream(dev.d, dev.rx, 0, 0);
            sx_device_deactivate_stream(dev.d, dev.tx, 0, 0);
            expect(sx_device_activate_stream(dev.d, dev.rx, 0, 0, 0) == 0, "re-activate rx");
            expect(sx_device_activate_stream(dev.d, dev.tx, 0, 0, 0) == 0, "re-activate tx");
            tx_accepted = 0;
            ++resets;
            continue;
        }
        if (op < 55) {
            // a read: mostly one period, sometimes larger (page-locked now and then), sometimes non-blocking
            // (the oracle-backed fake takes ~0.1 s for a large read and its read-ahead: rare, so that 10^5 calls stay minutes)
            size_t n = 256;
            float *dst = buf.data();
            const unsigned size_dice = (unsigned)(rnd() % 1000);
            if (size_dice < 4) { n = 33000 + rnd() % 30000; dst = static_cast<float *>(pin); }
            else if (size_dice < 30) n = 1 + rnd() % 8000;
            const long timeout = (op % 7 == 0) ? 0 : 100000;
            void *b[1] = {dst};
            int flags = 0;
            long long t_ns = 0;
            const int r = sx_device_read_stream(dev.d, dev.rx, b, n, &flags, &t_ns, timeout);
            if (r > 0) {
                const long long pos = sx_time_ns_to_ticks(t_ns, rate);
                if (!ref.holds(pos, (size_t)r, dst)) ++bad_blocks;
                ++reads;
            } else if (r < 0) {
                expect(false, "readStream error on the virtual clock");
            }
        } else if (op < 90) {
            // a write: in sequence, timed a few periods ahead, or timed in the past (discarded, reported as written)
            const size_t n = (op < 60) ? 1 + rnd() % 8000 : 256;
            const void *b[1] = {out.data()};
            int flags = 0;
            long long t_ns = 0;
            if (op >= 75) {
                flags = SX_SOAPY_SDR_HAS_TIME;
                const long long base = dev.setting("TX_POSITION");
                const long long target = op < 85 ? base + (long long)(rnd() % 2000) : base - 70000;
                t_ns = sx_ticks_to_time_ns(target > 0 ? target : 0, rate);
            }
            const long long before = dev.setting("TX_WRITTEN");
            const int r = sx_device_write_stream(dev.d, dev.tx, b, n, &flags, t_ns, 100000);
            expect(r >= 0, "writeStream error on the virtual clock");
            const long long after = dev.setting("TX_WRITTEN");
            expect(after == before || after == before + r, "TX_WRITTEN moves by the block or not at all (discarded)");
            tx_accepted += after - before;
            ++writes;
        } else {
            // the application was away: the clock runs ahead by up to two rings -> the next read skips (a chain jump)
            char v[32];
            std::snprintf(v, sizeof(v), "%llu", (unsigned long long)(rnd() % 140000));
            sx_device_write_setting(dev.d, "CLOCK_ADVANCE", v);
            ++skips;
        }
    }
    stop = true;
    third.join();
    expect(bad_blocks == 0, "every RX block is the oracle's at its timestamp's position");
    expect(dev.setting("TX_WRITTEN") == tx_accepted, "TX_WRITTEN accounts for every accepted block");
    std::printf("calls %ld reads %ld writes %ld clock_jumps %ld resets %ld bad %ld polls %ld\n", ncalls, reads, writes, skips, resets,
                bad_blocks, polls.load());
    sxfir_host_free(pin);
}

// What setSampleRate leaves in the register shadow for every supported rate (SoapySX.cpp:1192-1208: 0x12 bits 3-0 = clkout,
// 0x13 bit 7 = mant, bit 6 = m, bits 5-3 = n, RX and TX enabled again in 0x00): printed for the caller, which compares with
// tests/golden/rate_table.json = the reference's own sample_rates[] table.
static void scenario_rate_registers()
{
    for (double clock : {38.4e6, 32.0e6}) {
        sx_device *d = sx_device_make(clock == 38.4e6 ? "driver=sx,clock=virtual" : "driver=sx,clock=virtual,master_clock=32e6");
        expect(d != nullptr, "make");
        double rates[16];
        const int n = sx_device_list_sample_rates(d, SX_SOAPY_SDR_RX, 0, rates, 16);
        for (int i = 0; i < n; ++i) {
            expect(sx_device_set_sample_rate(d, SX_SOAPY_SDR_TX, 0, rates[i]) == 0, "set_sample_rate");
            unsigned r0 = 0, r12[2] = {0, 0};
            sx_device_read_registers(d, "", 0x00, &r0, 1);
            sx_device_read_registers(d, "", 0x12, r12, 2);
            std::printf("rate_regs %.0f %.17g %u %u %u\n", clock, rates[i], r12[0], r12[1], r0);
        }
        sx_device_unmake(d);
    }
}

// Where the reference is silent the module is silent (round 6): writeSetting has one key, "PA", three values and no else
// branch (SoapySX.cpp:1472-1493); readSetting is not overridden, so SoapySDR's default answers "" (:1495); setupStream
// ignores its one channel's list (:747) and checks lock, format, running, already-set-up in that order (:750-764).
static void scenario_boundary()
{
    sx_device *d = sx_device_make("driver=sx,clock=virtual");
    expect(d != nullptr, "make");
    char out[64] = "x";
    expect(sx_device_write_setting(d, "NO_SUCH_KEY", "1") == 0, "unknown key is ignored");
    expect(sx_device_write_setting(d, "PA", "SOMETIMES") == 0, "unknown PA value is ignored");
    expect(sx_device_read_setting(d, "PA", out, sizeof(out)) >= 0 && std::string(out) == "AUTO", "PA mode after construction");
    expect(sx_device_write_setting(d, "PA", "OFF") == 0 && sx_device_read_setting(d, "PA", out, sizeof(out)) >= 0 &&
               std::string(out) == "OFF", "PA OFF is remembered");
    std::strcpy(out, "x");
    expect(sx_device_read_setting(d, "NO_SUCH_KEY", out, sizeof(out)) >= 0 && out[0] == 0, "unknown key reads as the empty string");
    const size_t odd[2] = {5, 7};
    sx_stream *rx = sx_device_setup_stream(d, SX_SOAPY_SDR_RX, "CF32", odd, 2, "");
    expect(rx != nullptr, "the channel list of the one channel is ignored");
    expect(sx_device_setup_stream(d, SX_SOAPY_SDR_RX, "CS16", odd, 2, "") == nullptr &&
               std::strstr(sx_device_last_error(), "Only CF32") != nullptr, "format is checked before already-set-up");
    expect(sx_device_setup_stream(d, SX_SOAPY_SDR_RX, "CF32", odd, 1, "") == nullptr &&
               std::strstr(sx_device_last_error(), "setup already") != nullptr, "second setup of a direction");
    sx_device_close_stream(d, rx);
    sx_device_unmake(d);
    sx_device *d4 = sx_device_make("driver=sx,clock=virtual,channels=4");
    expect(d4 != nullptr, "make channels=4");
    const size_t one = 0;
    expect(sx_device_setup_stream(d4, SX_SOAPY_SDR_RX, "CS16", &one, 1, "") == nullptr &&
               std::strstr(sx_device_last_error(), "Only CF32") != nullptr, "format before the build's own list rule");
    expect(sx_device_setup_stream(d4, SX_SOAPY_SDR_RX, "CF32", &one, 1, "") == nullptr &&
               std::strstr(sx_device_last_error(), "all channels") != nullptr, "channels=N: the list names all or none");
    sx_device_unmake(d4);
}

// usage: device_probe [calls [blocks [strict|lenient]]]   lenient (the sanitizer builds, which run several times slower
// than the wall clock allows for): the RX thread may be skipped ahead by the overrun rule; the data checks stay
int main(int argc, char **argv)
{
    const long ncalls = argc > 1 ? std::atol(argv[1]) : 100000;
    const int nblk = argc > 2 ? std::atoi(argv[2]) : 40;
    const bool lenient = argc > 3 && std::string(argv[3]) == "lenient";
    sx_device_set_log_level(1);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("# %s: %.1f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    scenario_rate_registers();
    scenario_boundary();
    scenario_threads("threads", 4096, nblk, false, lenient);
    lap("threads");
    scenario_threads("megabyte", (size_t)1 << 18, 6, true, true);
    lap("megabyte");
    scenario_linked(3);
    lap("linked");
    scenario_calls(ncalls);
    lap("calls");
    std::printf("bad %d\n", bad.load());
    return bad.load() ? 1 : 0;
}
