// CPU-only probe of GpuChains.hpp (RxChain read-ahead / jump handling / channel layout, TxChain write-behind /
// silence / ring wrap) over the test-only fake backend (fake_sxfir.cpp).  Every sample a chain hands out or
// leaves in its sink is compared with one direct pass of the oracle over the same stream positions.
#include <cstdio>
#include <cstring>
#include <vector>

#include "GpuChains.hpp"

extern "C" {
#include "sx_oracle.h"
}
#include <atomic>
extern std::atomic<int> g_fake_launches;

static int bad = 0;
static void check(const char *what, const float *got, const float *want, size_t n_floats)
{
    size_t diff = 0;
    for (size_t i = 0; i < n_floats; ++i) diff += std::memcmp(got + i, want + i, 4) != 0;
    std::printf("%s %zu mismatches of %zu\n", what, diff, n_floats);
    bad += diff != 0;
}

static void rx_test()
{
    const int D = 8, NT = 256, NCH = 3;
    const uint64_t seed = 0x51255;
    const uint32_t first = 5;
    const size_t total = 200000;
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, D, 8.0, 1.0, taps.data());
    std::vector<std::vector<float>> ref(NCH, std::vector<float>(2 * total));
    for (int c = 0; c < NCH; ++c) {
        std::vector<float> x(2 * total * D);
        sxo_synth_iq(seed, first + c, 0, total * D, x.data());
        sxo_decim_f32(taps.data(), NT, D, 2, 4, x.data(), total * D, 0, total, ref[c].data());
    }
    sx::RxChain chain(0, D, 32, seed, first, NCH, false);
    int64_t pos = 0;
    const size_t sizes[] = {1, 255, 256, 256, 4095, 4097, 70000, 33, 8192, 1000};
    std::vector<std::vector<float>> buf(NCH);
    for (size_t n : sizes) {
        float *dsts[NCH];
        for (int c = 0; c < NCH; ++c) { buf[c].assign(2 * n, -1.0f); dsts[c] = buf[c].data(); }
        chain.produce(pos, n, dsts);
        for (int c = 0; c < NCH; ++c) check("rx_read", buf[c].data(), ref[c].data() + 2 * pos, 2 * n);
        pos += (int64_t)n;
    }
    const int launches_sequential = g_fake_launches.load();
    // a jump forward (overrun skip) and one backward (restart): the read-ahead is dropped, history re-primed
    for (int64_t jump : {(int64_t)150000, (int64_t)0, (int64_t)77}) {
        const size_t n = 3000;
        float *dsts[NCH];
        for (int c = 0; c < NCH; ++c) { buf[c].assign(2 * n, -1.0f); dsts[c] = buf[c].data(); }
        chain.produce(jump, n, dsts);
        for (int c = 0; c < NCH; ++c) check("rx_after_jump", buf[c].data(), ref[c].data() + 2 * jump, 2 * n);
    }
    // batching really happens: the 10 sequential reads above took far fewer GPU passes than samples / 256
    std::printf("rx_launches %d\n", launches_sequential);

    // Large reads into page-locked caller memory: the decimator stores straight into it.  One block holds all
    // channels at a uniform stride (what a numpy [nch, n] array registered with sxfir_host_register looks like);
    // the reads continue the stream where the staged batches end, then small reads continue after them.
    {
        const size_t n = 40000, stride = 40960;
        void *blk = nullptr;
        sxfir_host_alloc(&blk, 8 * stride * NCH);
        float *base = static_cast<float *>(blk);
        int64_t p = 5000;
        float *dsts[NCH];
        for (int c = 0; c < NCH; ++c) dsts[c] = base + 2 * stride * c;
        for (int rep = 0; rep < 2; ++rep) {
            std::memset(blk, 0xff, 8 * stride * NCH);
            chain.produce(p, n, dsts);
            for (int c = 0; c < NCH; ++c) check("rx_direct", dsts[c], ref[c].data() + 2 * p, 2 * n);
            p += (int64_t)n;
        }
        std::printf("rx_direct_samples %lld\n", (long long)chain.direct_samples());
        const size_t m = 700;
        for (int c = 0; c < NCH; ++c) { buf[c].assign(2 * m, -1.0f); dsts[c] = buf[c].data(); }
        chain.produce(p, m, dsts);
        for (int c = 0; c < NCH; ++c) check("rx_after_direct", buf[c].data(), ref[c].data() + 2 * p, 2 * m);
        // the same size into ordinary memory takes the staged path
        const long long before = (long long)chain.direct_samples();
        std::vector<std::vector<float>> big(NCH, std::vector<float>(2 * n));
        for (int c = 0; c < NCH; ++c) dsts[c] = big[c].data();
        p += (int64_t)m;
        chain.produce(p, n, dsts);
        for (int c = 0; c < NCH; ++c) check("rx_large_pageable", dsts[c], ref[c].data() + 2 * p, 2 * n);
        std::printf("rx_direct_unchanged %d\n", before == (long long)chain.direct_samples());
        sxfir_host_free(blk);
    }
}

// Passes of a megabyte and more: the decimated block lands in HBM and is DMA-copied to the host (staged batches
// and stores into page-locked caller memory alike); the copies to pageable memory are split over the copy pool.
static void rx_large_test()
{
    const int D = 4, NT = 128;
    const uint64_t seed = 0x51255;
    const size_t total = 2600000;
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, D, 8.0, 1.0, taps.data());
    std::vector<float> ref(2 * total);
    {
        std::vector<float> x(2 * total * D);
        sxo_synth_iq(seed, 0, 0, total * D, x.data());
        sxo_decim_f32(taps.data(), NT, D, 2, 4, x.data(), total * D, 0, total, ref.data());
    }
    sx::RxChain chain(0, D, 32, seed, 0, 1, false);
    int64_t pos = 0;
    std::vector<float> buf;
    for (size_t n : {(size_t)300000, (size_t)300000, (size_t)256, (size_t)524288}) {
        buf.assign(2 * n, -1.0f);
        float *dsts[1] = {buf.data()};
        chain.produce(pos, n, dsts);
        check("rx_large_read", buf.data(), ref.data() + 2 * pos, 2 * n);
        pos += (int64_t)n;
    }
    void *blk = nullptr;
    const size_t n = 200000;
    sxfir_host_alloc(&blk, 8 * n);
    float *dsts[1] = {static_cast<float *>(blk)};
    pos += 1000;                                           // a jump: nothing of this read is staged
    std::memset(blk, 0xff, 8 * n);
    chain.produce(pos, n, dsts);
    check("rx_large_direct", dsts[0], ref.data() + 2 * pos, 2 * n);
    std::printf("rx_large_direct_samples %lld\n", (long long)chain.direct_samples());
    // the reader of page-locked megabyte blocks goes on: its read-ahead stays in HBM and is DMA-copied into its buffer
    pos += (int64_t)n;
    for (int rep = 0; rep < 3; ++rep) {
        std::memset(blk, 0xff, 8 * n);
        chain.produce(pos, n, dsts);
        check("rx_large_from_hbm", dsts[0], ref.data() + 2 * pos, 2 * n);
        pos += (int64_t)n;
    }
    std::printf("rx_large_hbm_samples %lld\n", (long long)chain.direct_samples());
    // ... and an ordinary buffer after all: the batch in HBM takes the staging hop
    buf.assign(2 * 300000, -1.0f);
    float *pd[1] = {buf.data()};
    chain.produce(pos, 300000, pd);
    check("rx_large_fallback", buf.data(), ref.data() + 2 * pos, 2 * 300000);
    pos += 300000;
    chain.produce(pos, 1000, pd);
    check("rx_large_small_after", buf.data(), ref.data() + 2 * pos, 2 * 1000);
    sxfir_host_free(blk);
}

// Large writes: the slots grow, their samples reach HBM by DMA copies; page-locked caller memory is taken as it is.
static void tx_large_test()
{
    const int L = 8, NT = 256;
    const size_t ring_frames = 1 << 16;
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, L, 8.0, (double)L, taps.data());
    sx::TxChain chain(0, L, 32, ring_frames, 1, false);
    chain.set_threshold2(0.5f);
    const size_t total = 900000;
    std::vector<float> stream(2 * total, 0.0f);
    sxo_synth_iq(123, 3, 0, total, stream.data());
    // a gap of silence in the middle (a timed write further on)
    std::memset(stream.data() + 2 * 600100, 0, 8 * 900);
    long long want_keyed = 0;
    auto count = [&](size_t from, size_t n) {
        for (size_t i = from; i < from + n; ++i)
            want_keyed += (stream[2 * i] * stream[2 * i] + stream[2 * i + 1] * stream[2 * i + 1] >= 0.5f) ? 1 : 0;
    };
    const float *srcs[1];
    srcs[0] = stream.data();               chain.consume(0, 300, srcs);          count(0, 300);
    srcs[0] = stream.data() + 2 * 300;     chain.consume(300, 299800, srcs);     count(300, 299800);      // pageable, large
    std::printf("tx_large_slot_frames %zu\n", chain.slot_frames());
    srcs[0] = stream.data() + 2 * 300100;  chain.consume(300100, 300000, srcs);  count(300100, 300000);
    void *blk = nullptr;
    const size_t nd = 150000;
    sxfir_host_alloc(&blk, 8 * nd);
    std::memcpy(blk, stream.data() + 2 * 601000, 8 * nd);
    srcs[0] = static_cast<const float *>(blk);
    chain.consume(601000, nd, srcs);                                              // after the gap, from page-locked memory
    count(601000, nd);
    std::memset(blk, 0x55, 8 * nd);                        // the call has returned: the memory is the caller's again
    std::printf("tx_large_direct_samples %lld\n", (long long)chain.direct_samples());
    srcs[0] = stream.data() + 2 * 751000;  chain.consume(751000, 149000, srcs);  count(751000, 149000);
    std::printf("tx_large_keyed %lld want %lld\n", (long long)chain.keyed_samples(), want_keyed);
    bad += chain.keyed_samples() != want_keyed;
    std::vector<float> want(2 * total * L), got(2 * ring_frames * L);
    sxo_interp_f32(taps.data(), NT, L, 2, stream.data(), total, 0, total * L, want.data());
    const int64_t from = (int64_t)total - (int64_t)ring_frames;
    chain.capture(from * L, ring_frames * L, got.data(), 0);
    check("tx_large_sink", got.data(), want.data() + 2 * from * L, 2 * ring_frames * L);
    sxfir_host_free(blk);
}

// Random call sequences: block sizes from one sample to megabytes, ordinary and page-locked buffers, jumps forwards
// and backwards, in any order -- every path change of the chains (staged / in HBM / direct, zero-copy / DMA, slot
// growth) must keep the stream intact.
static uint64_t rnd_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd()
{
    rnd_state ^= rnd_state << 13;
    rnd_state ^= rnd_state >> 7;
    rnd_state ^= rnd_state << 17;
    return rnd_state;
}

static size_t random_block()
{
    switch (rnd() % 6) {
    case 0: return 1 + rnd() % 300;
    case 1: return 256 * (1 + rnd() % 16);
    case 2: return 30000 + rnd() % 10000;
    case 3: return 131072 + rnd() % 70000;
    case 4: return 262144 + rnd() % 300000;
    default: return 1 + rnd() % 5000;
    }
}

static void rx_random_test()
{
    const int D = 4, NT = 128;
    const uint64_t seed = 77;
    const size_t total = 6000000;
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, D, 8.0, 1.0, taps.data());
    std::vector<float> ref(2 * total);
    {
        std::vector<float> x(2 * total * D);
        sxo_synth_iq(seed, 2, 0, total * D, x.data());
        sxo_decim_f32(taps.data(), NT, D, 2, 4, x.data(), total * D, 0, total, ref.data());
    }
    sx::RxChain chain(0, D, 32, seed, 2, 1, false);
    void *blk = nullptr;
    sxfir_host_alloc(&blk, 8 * 600000);
    std::vector<float> plain(2 * 600000);
    int64_t pos = 0;
    size_t bad_reads = 0, reads = 0;
    for (int op = 0; op < 90; ++op) {
        const size_t n = random_block();
        if (rnd() % 7 == 0) pos = (int64_t)(rnd() % (total - 700000));     // a jump, either way
        if (pos + (int64_t)n > (int64_t)total) pos = 0;
        const bool locked = rnd() % 2 == 0;
        float *dst = locked ? static_cast<float *>(blk) : plain.data();
        std::memset(dst, 0xff, 8 * n);
        float *dsts[1] = {dst};
        chain.produce(pos, n, dsts);
        bad_reads += std::memcmp(dst, ref.data() + 2 * pos, 8 * n) != 0;
        ++reads;
        pos += (int64_t)n;
    }
    std::printf("rx_random %zu mismatches of %zu reads\n", bad_reads, reads);
    bad += bad_reads != 0;
    sxfir_host_free(blk);
}

static void tx_random_test()
{
    const int L = 4, NT = 128;
    const size_t ring_frames = 1 << 20;
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, L, 8.0, (double)L, taps.data());
    sx::TxChain chain(0, L, 32, ring_frames, 1, false);
    chain.set_threshold2(0.3f);
    const size_t cap = 9000000;
    std::vector<float> stream(2 * cap, 0.0f);
    void *blk = nullptr;
    sxfir_host_alloc(&blk, 8 * 600000);
    std::vector<float> plain(2 * 600000);
    int64_t pos = 0;
    long long want_keyed = 0;
    for (int op = 0; op < 70; ++op) {
        const size_t n = random_block();
        if (rnd() % 5 == 0) pos += (int64_t)(rnd() % 3000);               // a gap: silence
        if (pos + (int64_t)n > (int64_t)cap) break;
        const bool locked = rnd() % 2 == 0;
        float *src = locked ? static_cast<float *>(blk) : plain.data();
        sxo_synth_iq(5, 9, pos, n, src);
        std::memcpy(stream.data() + 2 * pos, src, 8 * n);
        for (size_t i = 0; i < n; ++i) want_keyed += (src[2 * i] * src[2 * i] + src[2 * i + 1] * src[2 * i + 1] >= 0.3f) ? 1 : 0;
        const float *srcs[1] = {src};
        chain.consume(pos, n, srcs);
        std::memset(src, 0x33, 8 * n);                     // the call has returned: the memory is the caller's again
        pos += (int64_t)n;
    }
    const size_t total = (size_t)pos;
    std::printf("tx_random_keyed %lld want %lld\n", (long long)chain.keyed_samples(), want_keyed);
    bad += chain.keyed_samples() != want_keyed;
    const size_t tail = std::min(total, ring_frames);
    std::vector<float> want(2 * total * L), got(2 * tail * L);
    sxo_interp_f32(taps.data(), NT, L, 2, stream.data(), total, 0, total * L, want.data());
    chain.capture((int64_t)(total - tail) * L, tail * L, got.data(), 0);
    check("tx_random_sink", got.data(), want.data() + 2 * (total - tail) * L, 2 * tail * L);
    sxfir_host_free(blk);
}

static void tx_test()
{
    const int L = 8, NT = 256, NCH = 2;
    const size_t ring_frames = 1 << 14;                    // sink ring of 2^14 * L DAC samples per channel
    std::vector<float> taps(NT);
    sxo_design_lowpass(NT, L, 8.0, (double)L, taps.data());
    sx::TxChain chain(0, L, 32, ring_frames, NCH, false);
    // stream: blocks with gaps (timed writes / underrun skips leave silence), one block longer than a slot
    struct Blk { int64_t pos; size_t n; };
    const Blk blocks[] = {{0, 256}, {256, 256}, {1000, 100}, {1100, 40000}, {50000, 3000}, {53000, 1}, {60000, 5000}};
    const size_t total = 65000;
    std::vector<std::vector<float>> stream(NCH, std::vector<float>(2 * total, 0.0f));
    for (const Blk &b : blocks) {
        const float *srcs[NCH];
        std::vector<std::vector<float>> tmp(NCH, std::vector<float>(2 * b.n));
        for (int c = 0; c < NCH; ++c) {
            sxo_synth_iq(99, 10 + c, b.pos, b.n, tmp[c].data());
            std::memcpy(stream[c].data() + 2 * b.pos, tmp[c].data(), 8 * b.n);
            srcs[c] = tmp[c].data();
        }
        chain.consume(b.pos, b.n, srcs);
    }
    std::printf("tx_written %lld\n", (long long)chain.written());
    // keying count (SoapySX.cpp:132-133) of channel 0's application samples, silence excluded, as the GPU pass counts it
    {
        chain.set_threshold2(0.25f);
        // (the threshold applies to blocks flushed from now on; count what a fresh chain sees instead)
        sx::TxChain kc(0, L, 32, ring_frames, NCH, false);
        kc.set_threshold2(0.25f);
        long long want = 0;
        for (const Blk &b : blocks) {
            std::vector<std::vector<float>> tmp(NCH, std::vector<float>(2 * b.n));
            const float *srcs[NCH];
            for (int c = 0; c < NCH; ++c) { sxo_synth_iq(99, 10 + c, b.pos, b.n, tmp[c].data()); srcs[c] = tmp[c].data(); }
            for (size_t i = 0; i < b.n; ++i) {
                const float ii = tmp[0][2 * i] * tmp[0][2 * i], qq = tmp[0][2 * i + 1] * tmp[0][2 * i + 1];
                want += (ii + qq >= 0.25f) ? 1 : 0;
            }
            kc.consume(b.pos, b.n, srcs);
        }
        std::printf("tx_keyed %lld want %lld\n", (long long)kc.keyed_samples(), want);
        bad += kc.keyed_samples() != want;
        kc.reset();
        std::printf("tx_keyed_after_reset %lld\n", (long long)kc.keyed_samples());
    }
    // the sink holds the last ring_frames stream samples' worth of output: compare the retained tail
    const int64_t end = 65000;
    const int64_t from = end - (int64_t)ring_frames;
    for (int c = 0; c < NCH; ++c) {
        std::vector<float> want(2 * total * L);
        sxo_interp_f32(taps.data(), NT, L, 2, stream[c].data(), total, 0, total * L, want.data());
        std::vector<float> got(2 * ring_frames * L);
        chain.capture(from * L, ring_frames * L, got.data(), c);
        check("tx_sink", got.data(), want.data() + 2 * from * L, 2 * ring_frames * L);
    }
    // positions cannot move backwards
    try {
        const float z[2] = {0, 0};
        const float *srcs[NCH] = {z, z};
        chain.consume(10, 1, srcs);
        std::printf("tx_backwards accepted\n");
    } catch (const std::exception &e) {
        std::printf("tx_backwards refused\n");
    }
    // after a reset the stream restarts at 0 with a clean history
    chain.reset();
    std::vector<float> blk(2 * 512), want(2 * 512 * L), got(2 * 512 * L);
    sxo_synth_iq(7, 1, 0, 512, blk.data());
    const float *srcs[NCH] = {blk.data(), blk.data()};
    chain.consume(0, 512, srcs);
    sxo_interp_f32(taps.data(), NT, L, 2, blk.data(), 512, 0, 512 * L, want.data());
    chain.capture(0, 512 * L, got.data(), 1);
    check("tx_after_reset", got.data(), want.data(), 2 * 512 * L);
}

int main()
{
    rx_test();
    tx_test();
    rx_large_test();
    tx_large_test();
    rx_random_test();
    tx_random_test();
    std::printf("bad %d\n", bad);
    return bad ? 1 : 0;
}
