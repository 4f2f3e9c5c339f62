// CPU sweep of StreamRules.hpp (what SoapySXHip::readStream / writeStream decide before any sample moves)
// against the oracle's independent restatement of SoapySX.cpp:897-1100 (sxo_rx_step / sxo_tx_step): random and
// edge-case counters, every decision compared.  Prints "cases N mismatches M".
#include <cstdint>
#include <cstdio>
#include <random>

#include "StreamRules.hpp"

extern "C" {
#include "sx_oracle.h"
}

int main()
{
    std::mt19937_64 rng(0x51255);
    auto pick = [&](int64_t lo, int64_t hi) { return lo + (int64_t)(rng() % (uint64_t)(hi - lo + 1)); };
    const uint64_t periods[] = {256, 1000, 4096, 65536};
    const double rates[] = {600000.0, 300000.0, 75000.0, 38.4e6 / 1536, 32.0e6 / 768};
    long cases = 0, bad = 0;
    for (int it = 0; it < 200000; ++it) {
        const uint64_t period = periods[rng() % 4];
        const uint64_t ring = 65536 / period * period;
        const double rate = rates[rng() % 5];
        const int64_t position = pick(0, 1) ? pick(0, 5000000) : pick(0, 4000000000000ll);
        const size_t want = (size_t)(pick(0, 9) == 0 ? pick(0, 3) : pick(1, 200000));
        const long timeout = pick(0, 2) == 0 ? 0 : (pick(0, 5) == 0 ? -1 : 100000);

        // ---- capture: avail around the ring size, far beyond it, zero and negative (xrun-ish) values
        {
            int64_t avail;
            switch (rng() % 5) {
            case 0: avail = pick(-10, 10); break;
            case 1: avail = (int64_t)ring + pick(-3, 3); break;
            case 2: avail = (int64_t)ring + pick(1, 5) * (int64_t)period + pick(-2, 2); break;
            case 3: avail = pick(0, 3000000); break;
            default: avail = pick(0, (int64_t)ring); break;
            }
            sxo_stream_result r;
            sxo_rx_step(position, avail, period, ring, want, timeout, rate, &r);
            int64_t skip = sx::rules::rx_overrun_skip(avail, ring, period);
            // what the PCM does with the rule's request: snd_pcm_forward cannot move past what is available
            if (skip > 0 && skip > avail) skip = avail;
            const uint64_t len = sx::rules::request_length(want, avail - skip, timeout);
            const int64_t end = position + skip + (int64_t)len;
            ++cases;
            if (skip != r.skipped || (int64_t)len != r.length || end != r.position) {
                if (bad++ < 5) std::printf("rx mismatch: avail %lld ring %llu period %llu want %zu timeout %ld -> skip %lld/%lld len %llu/%lld\n",
                                           (long long)avail, (unsigned long long)ring, (unsigned long long)period, want, timeout,
                                           (long long)skip, (long long)r.skipped, (unsigned long long)len, (long long)r.length);
            }
        }
        // ---- playback: delay below, at and above zero (underrun = negative delay), timestamps around "now"
        {
            const int64_t delay = pick(0, 3) == 0 ? -pick(0, 300000) : pick(0, (int64_t)ring);
            const int64_t avail = (int64_t)ring - (delay > 0 ? delay : 0) + (pick(0, 7) == 0 ? -pick(0, 70000) : 0);
            const bool timed = pick(0, 1) == 1;
            const int64_t playing = position - delay;
            const int64_t target = playing + (pick(0, 2) == 0 ? pick(-5, 5) : pick(-200000, 400000));
            const long long time_ns = sxo_ticks_to_time_ns(target < 0 ? 0 : target, rate);
            sxo_stream_result r;
            sxo_tx_step(position, avail, delay, period, want, timed ? 4 /* SOAPY_SDR_HAS_TIME */ : 0, time_ns, timeout, rate, &r);
            const sx::rules::TxPlacement p = sx::rules::tx_placement(position, delay, period, timed,
                                                                    timed ? sxo_time_ns_to_ticks(time_ns, rate) : 0);
            ++cases;
            const bool dropped = p.kind == sx::rules::TxPlacement::IN_THE_PAST;
            bool ok = dropped == (r.discarded != 0);
            if (ok && dropped) ok = r.ret == (int)want && r.position == position;
            if (ok && !dropped) {
                const int64_t gap = p.write_position - position > 0 ? p.write_position - position : 0;
                const uint64_t len = sx::rules::request_length(want, avail - gap, timeout);
                ok = gap == r.skipped && (int64_t)len == r.length && position + gap + (int64_t)len == r.position;
                ok = ok && ((p.kind == sx::rules::TxPlacement::PAST_UNDERRUN) == (!timed && playing > position));
            }
            if (!ok && bad++ < 5)
                std::printf("tx mismatch: pos %lld delay %lld avail %lld period %llu timed %d target %lld want %zu timeout %ld kind %d\n",
                            (long long)position, (long long)delay, (long long)avail, (unsigned long long)period, (int)timed,
                            (long long)target, want, timeout, (int)p.kind);
        }
    }
    std::printf("cases %ld mismatches %ld\n", cases, bad);
    return bad ? 1 : 0;
}
