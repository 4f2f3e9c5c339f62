// CPU-only probe of the host logic behind the Device (no GPU, no HIP): the sample-clock / PCM model
// (SynthPcm.hpp, the counterpart of AlsaPcm, SoapySX.cpp:369-518) and the SX1255 register shadow
// (Sx1255Shadow.hpp, SoapySX.cpp:1225-1561).  Prints "key value..." lines that
// tests/test_host_logic.py compares with hand-evaluated ALSA semantics and with the oracle.
#include <cstdio>
#include <string>
#include <vector>

#include "SynthPcm.hpp"
#include "Sx1255Shadow.hpp"
#include <SoapySDR/Time.hpp>

using namespace sx;

static void geometry()
{
    SampleClock clk(SampleClock::VIRTUAL, 75000.0);
    SynthPcm rx("rx", SynthPcm::CAPTURE, &clk);
    for (unsigned long period : {0ul, 256ul, 1000ul, 65536ul, 100000ul}) {
        rx.configure(period);
        std::printf("geometry %lu %llu %llu\n", period, (unsigned long long)rx.hwp_period_size,
                    (unsigned long long)rx.hwp_buffer_size);
    }
}

static void capture_normal()
{
    SampleClock clk(SampleClock::VIRTUAL, 75000.0);
    SynthPcm rx("rx", SynthPcm::CAPTURE, &clk);
    rx.configure(256);
    int64_t avail = -1, delay = -1, first = -1;
    rx.start();
    clk.advance(1000);
    int rc = rx.avail_delay(&avail, &delay);
    std::printf("capture_after_1000 %d %lld %lld\n", rc, (long long)avail, (long long)delay);
    int64_t got = rx.begin_read(256, &first);                     // data is there: no waiting
    std::printf("capture_read %lld %lld %lld\n", (long long)got, (long long)first, (long long)clk.now());
    got = rx.begin_read(2000, &first);                            // blocks: the virtual clock runs to cover it
    std::printf("capture_blocking_read %lld %lld %lld\n", (long long)got, (long long)first, (long long)clk.now());
    clk.advance(70000);                                           // NORMAL mode never stops on overrun
    rc = rx.avail_delay(&avail, &delay);
    std::printf("capture_overrun_normal %d %lld %d\n", rc, (long long)avail, (int)(rx.state() == SynthPcm::RUNNING));
    const int64_t moved = rx.forward(5000);
    std::printf("capture_forward %lld %lld\n", (long long)moved, (long long)rx.appl());
}

static void playback_and_link()
{
    SampleClock clk(SampleClock::VIRTUAL, 75000.0);
    SynthPcm rx("rx", SynthPcm::CAPTURE, &clk), tx("tx", SynthPcm::PLAYBACK, &clk);
    rx.configure(256);
    tx.configure(256);
    rx.stream_mode = tx.stream_mode = STREAM_MODE_LINK;
    rx.link(&tx);
    int64_t avail = -1, delay = -1, first = -1;
    // prepared, not running: a playback ring only takes what fits
    int64_t w = tx.begin_write(70000, &first);
    std::printf("link_first_write %lld %lld %d %d\n", (long long)w, (long long)first, (int)(tx.state() == SynthPcm::RUNNING),
                (int)(rx.state() == SynthPcm::RUNNING));          // the first write starts both (start_threshold 1)
    clk.advance(1000);
    tx.avail_delay(&avail, &delay);
    std::printf("link_tx_after_1000 %lld %lld\n", (long long)avail, (long long)delay);
    rx.avail_delay(&avail, &delay);
    std::printf("link_rx_after_1000 %lld %lld\n", (long long)avail, (long long)delay);
    clk.advance(64536);                                           // playback runs dry exactly now: hw == appl
    int rc = tx.avail_delay(&avail, &delay);
    std::printf("link_underrun %d %d %d\n", rc, (int)(tx.state() == SynthPcm::XRUN), (int)(rx.state() == SynthPcm::XRUN));
    std::printf("link_rx_after_xrun %d\n", rx.avail_delay(&avail, &delay));
    tx.reset();
    rx.reset();
    std::printf("link_after_reset %d %lld %lld\n", (int)(tx.state() == SynthPcm::PREPARED), (long long)tx.appl(),
                (long long)tx.position);
}

static void shadow()
{
    for (double clock : {38.4e6, 32.0e6}) {
        Sx1255Shadow chip(clock);
        for (double f : {433.92e6, 434.0e6, 0.0, 1.0e12, 144.39e6}) {
            chip.tune(true, f);
            const auto r = chip.read(0x01, 3);
            std::printf("tune %.17g %.17g %.17g %u\n", clock, f, chip.tuned(true), (r[0] << 16) | (r[1] << 8) | r[2]);
        }
        for (double g : {-5.0, 0.0, 7.0, 12.0, 30.0, 41.0, 47.9, 60.0, 78.0, 100.0}) {
            chip.set_overall_gain(true, g);
            std::printf("rxgain %.1f %.1f %.1f %u\n", g, chip.gain(true, "LNA"), chip.gain(true, "PGA"), chip.read(0x0C, 1)[0]);
            chip.set_overall_gain(false, g);
            std::printf("txgain %.1f %.1f %.1f %u\n", g, chip.gain(false, "DAC"), chip.gain(false, "MIXER"), chip.read(0x08, 1)[0]);
        }
    }
    Sx1255Shadow chip(38.4e6);
    std::printf("boot");
    for (unsigned v : chip.read(0, 0x14)) std::printf(" %u", v);
    std::printf("\n");
    chip.set_antenna(true, "DLB");
    chip.set_antenna(false, "NONE");
    std::printf("antenna %s %s %u %u\n", chip.antenna(true).c_str(), chip.antenna(false).c_str(), chip.read(0x10, 1)[0],
                chip.read(0x00, 1)[0]);
    try {
        chip.write(0x7F, {1, 2});
        std::printf("burst_over_end accepted\n");
    } catch (const std::exception &e) {
        std::printf("burst_over_end %s\n", e.what());
    }
}

static void ticks()
{
    for (double rate : {600000.0, 300000.0, 75000.0, 38.4e6 / 1536, 32.0e6 / 768}) {
        for (long long t : {0ll, 256ll, 768ll, 1000000007ll, 123456789012ll})
            std::printf("ticks %.17g %lld %lld %lld\n", rate, t, (long long)SoapySDR::ticksToTimeNs(t, rate),
                        (long long)SoapySDR::timeNsToTicks(SoapySDR::ticksToTimeNs(t, rate), rate));
    }
}

int main()
{
    geometry();
    capture_normal();
    playback_and_link();
    shadow();
    ticks();
    return 0;
}
