"""The SoapySDR ``driver=sx`` surface on the GPU box, exercised the way the
reference's own scripts exercise it (SoapySX/test/test_timestamps.py,
test_linked_streams.py, example/linear_repeater.py), but asserting instead of
printing.  A virtual sample clock (device arg clock=virtual) makes every run
deterministic; sample data is checked bit for bit against the oracle and
position / timestamp arithmetic against the oracle's restatement of
SoapySX.cpp:897-1104."""
import os

import numpy as np
import pytest

import sxxcvr_amd
import sxxcvr_amd.soapy as SoapySDR
from gpu_util import assert_bit_exact

pytestmark = pytest.mark.gpu

SEED = 0x51255
RATE = 75000.0


def make(**extra):
    args = {"driver": "sx", "clock": "virtual"}
    args.update(extra)
    dev = SoapySDR.Device(args)
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, RATE)
    dev.setSampleRate(SoapySDR.SOAPY_SDR_TX, 0, RATE)
    return dev


def rx_reference(oracle, decim, n_out):
    """The stream the RX chain must deliver: decimated synthetic source, from stream position 0."""
    h = sxxcvr_amd.design_lowpass(32 * decim, decim)
    x = oracle.synth_iq(SEED, 0, 0, n_out * decim)
    return oracle.decim_f32(h, decim, x, 2, 4)


def tx_reference(oracle, interp, stream):
    h = sxxcvr_amd.design_lowpass(32 * interp, interp, 8.0, float(interp))
    return oracle.interp_f32(h, interp, stream, 2)


def test_identity_and_formats():
    dev = make()
    assert dev.getDriverKey() == "sx" and dev.getHardwareKey() == "sx"      # SoapySX.cpp:1567-1575
    info = dev.getHardwareInfo()
    assert info["hardware_version"] == "unknown" and "soapysx_tag" in info and info["gpu_arch"] == "gfx950"
    assert dev.getNumChannels(SoapySDR.SOAPY_SDR_RX) == 1 and dev.getNumChannels(SoapySDR.SOAPY_SDR_TX) == 1
    assert dev.getStreamFormats(SoapySDR.SOAPY_SDR_RX, 0) == ["CF32"]
    assert dev.getNativeStreamFormat(SoapySDR.SOAPY_SDR_TX, 0) == ("CF32", 1.0)
    assert dev.hasHardwareTime("") and not dev.hasHardwareTime("pps")
    with pytest.raises(RuntimeError, match="Unsupported time"):
        dev.getHardwareTime("pps")


def test_sample_rate_table():
    dev = make()
    rates = dev.listSampleRates(SoapySDR.SOAPY_SDR_RX, 0)
    assert rates == [38.4e6 / d for d in (1536, 768, 512, 256, 128, 64)]     # SoapySX.cpp:196-208
    dev32 = SoapySDR.Device({"driver": "sx", "clock": "virtual", "master_clock": "32e6"})
    assert dev32.listSampleRates(SoapySDR.SOAPY_SDR_TX, 0) == [32e6 / d for d in (1536, 768, 512, 256, 128, 64)]
    assert dev32.getSampleRate(SoapySDR.SOAPY_SDR_RX, 0) == 125000.0          # masterClock / 256, :662
    import json
    table = {row["div"]: row for row in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rate_table.json")))["rows"]}
    for r in rates:
        dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, r)
        assert dev.getSampleRate(SoapySDR.SOAPY_SDR_RX, 0) == r
        # the register shadow holds what the reference programs for this rate (:1197-1203), per its own table
        row = table[int(round(38.4e6 / r))]
        r12, r13 = dev.readRegisters("", 0x12, 2)
        assert r12 & 0x0F == row["clkout"] and (r13 >> 7) & 1 == row["mant"] and (r13 >> 6) & 1 == row["m"] and (r13 >> 3) & 7 == row["n"]
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 75010.0)                      # rounds to the nearest divider, :1179
    assert dev.getSampleRate(SoapySDR.SOAPY_SDR_TX, 0) == 75000.0
    for bad in (100000.0, 38.4e6 / 384, 1.0):
        with pytest.raises(RuntimeError, match="Unsupported sample rate"):
            dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, bad)
    for bad in (0.0, -75000.0, float("nan")):
        with pytest.raises(RuntimeError, match="Sample rate must be positive"):
            dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, bad)


def test_stream_lifecycle_errors():
    dev = make()
    with pytest.raises(RuntimeError, match="Only CF32"):
        dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CS16", [0], {})
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    with pytest.raises(RuntimeError, match="setup already"):
        dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "1000"})
    assert dev.getStreamMTU(rx) == 256 and dev.getStreamMTU(tx) == 1000       # :451, :861-866
    buf = np.zeros(256, dtype=np.complex64)
    # not activated -> 0, no flags (:893-894, :984-985)
    r = dev.readStream(rx, [buf], 256)
    assert (r.ret, r.flags) == (0, 0)
    assert dev.writeStream(tx, [buf], 256).ret == 0
    with pytest.raises(RuntimeError, match="Wrong direction"):
        dev.readStream(tx, [buf], 256)
    with pytest.raises(RuntimeError, match="Wrong direction"):
        dev.writeStream(rx, [buf], 256)
    assert dev.activateStream(rx) == 0
    assert dev.activateStream(rx) == SoapySDR.SOAPY_SDR_STREAM_ERROR          # :815-818
    dev.closeStream(tx)
    with pytest.raises(RuntimeError, match="none of the streams are running"):
        dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})   # :754-758
    assert dev.deactivateStream(rx) == 0
    assert dev.deactivateStream(rx) == SoapySDR.SOAPY_SDR_STREAM_ERROR        # :843-846


def test_outer_boundary_behaves_like_the_reference_where_it_is_silent():
    """Round 6: three places where the module used to be stricter than SoapySX.cpp, now the reference's behaviour.
    writeSetting has one key, "PA", with three values, and no else branch (:1472-1493): anything else is ignored.
    The reference does not override readSetting (":1495 TODO"), so SoapySDR's default answers "" for every key.
    setupStream ignores the channel list of its one channel (:747) and checks in the order lock, format, running,
    already set up (:750-764)."""
    dev = make()
    dev.writeSetting("NO_SUCH_KEY", "1")                       # ignored, no exception
    dev.writeSetting("PA", "SOMETIMES")                        # an unknown PA value: ignored, the mode stays
    assert dev.readSetting("PA") == "AUTO"                     # both GPIO lines high after construction (:685-696) = AUTO
    for mode in ("ON", "OFF", "AUTO"):
        dev.writeSetting("PA", mode)
        assert dev.readSetting("PA") == mode
    assert dev.readSetting("NO_SUCH_KEY") == ""
    assert dev.readSetting("") == ""
    assert int(dev.readSetting("RX_DECIM")) >= 4               # the build's own keys keep working
    # the list is ignored on a one-channel device: any content sets the stream up
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [5, 7], {})
    # order of the checks: a wrong format is reported before "already set up", and both before the list is looked at
    with pytest.raises(RuntimeError, match="Only CF32"):
        dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CS16", [5, 7], {})
    with pytest.raises(RuntimeError, match="setup already"):
        dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [9], {})
    dev.closeStream(rx)
    dev4 = make(channels="4")
    with pytest.raises(RuntimeError, match="Only CF32"):       # format before the (build-defined) list rule
        dev4.setupStream(SoapySDR.SOAPY_SDR_RX, "CS16", [0], {})
    with pytest.raises(RuntimeError, match="all channels"):
        dev4.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    # the commit the module reports is the tree's (sxxcvr_amd/build.py stamps `git rev-parse HEAD`), not a constant
    commit = dev.getHardwareInfo()["soapysx_commit"]
    assert commit != "round1" and (commit == "unknown" or len(commit.split("-")[0]) == 40), commit


def test_rx_timestamps_and_data(oracle):
    """SoapySX/test/test_timestamps.py: untimed reads of one period; time = position / rate."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    dev.activateStream(tx)
    ref = rx_reference(oracle, 4, 256 * 12)
    buf = np.zeros(256, dtype=np.complex64)
    times = []
    for i in range(12):
        before = dev.getHardwareTime()
        r = dev.readStream(rx, [buf], len(buf))
        after = dev.getHardwareTime()
        assert r.ret == 256 and r.flags == SoapySDR.SOAPY_SDR_HAS_TIME
        assert r.timeNs == oracle.ticks_to_time_ns(256 * i, RATE)
        assert_bit_exact(buf, ref[256 * i:256 * (i + 1)], "rx block %d" % i)
        # delay from the last RX sample to "now": at most one sample period on the virtual clock
        d = after - (r.timeNs + int(round(1.0e9 * (r.ret - 1) / RATE)))
        assert 0 <= d <= int(1e9 / RATE) + 1 and before <= after
        times.append(r.timeNs)
    assert times[:3] == [0, 3413333, 6826667]
    # reads that are not period-aligned keep sample-exact positions
    odd = np.zeros(100, dtype=np.complex64)
    r = dev.readStream(rx, [odd], 100)
    assert r.ret == 100 and r.timeNs == oracle.ticks_to_time_ns(256 * 12, RATE)
    assert dev.readSetting("RX_POSITION") == str(256 * 12 + 100)


def test_full_duplex_timed_loop(oracle):
    """example/linear_repeater.py:40-69: every RX block is retransmitted with a timestamp
    768 samples later; the TX block must land at exactly rx_position + 768."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    latency = 768
    dt = int(round(latency * 1e9 / RATE))
    assert dt == 10240000
    nblk = 10
    buf = np.zeros(256, dtype=np.complex64)
    stream = np.zeros(latency + 256 * nblk, dtype=np.complex64)
    for i in range(nblk):
        r = dev.readStream(rx, [buf], len(buf))
        assert r.ret == 256
        t = dev.writeStream(tx, [buf], len(buf), flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=r.timeNs + dt)
        assert t.ret == 256
        assert int(dev.readSetting("TX_POSITION")) == 256 * i + latency + 256
        stream[latency + 256 * i: latency + 256 * (i + 1)] = buf
    assert int(dev.readSetting("TX_WRITTEN")) == 256 * nblk
    assert int(dev.readSetting("TX_PTT_SAMPLES")) == 256 * nblk        # threshold 0 keeps the PA keyed
    # what reached the synthetic DAC: the interpolated stream, silence before the first block
    L = int(dev.readSetting("TX_INTERP"))
    got = dev.txCapture(0, len(stream) * L)
    assert_bit_exact(got, tx_reference(oracle, L, stream), "dac stream")
    assert not got[: (latency - 40) * L].any()


def test_rx_overrun_skip(oracle):
    """RX overrun: more than the ring is pending -> whole periods + margin are skipped (:910-927)."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    buf = np.zeros(256, dtype=np.complex64)
    assert dev.readStream(rx, [buf], 256).ret == 256
    dev.writeSetting("CLOCK_ADVANCE", 70000)                           # application stalls
    pos, avail = 256, 70000
    exp = oracle.rx_step(pos, avail, 256, 65536, 256, 100000, RATE)
    SoapySDR.drainLog()
    r = dev.readStream(rx, [buf], 256)
    assert (r.ret, r.flags, r.timeNs) == (exp.ret, exp.flags, exp.time_ns)
    assert exp.skipped == ((70000 - 65536) // 256 + 2) * 256
    assert int(dev.readSetting("RX_POSITION")) == exp.position
    assert "RX buffer overrun. Skipped %d samples" % exp.skipped in SoapySDR.drainLog()
    ref = rx_reference(oracle, 4, exp.position)
    assert_bit_exact(buf, ref[exp.position - 256: exp.position], "data after the skip")


def test_rx_nonblocking(oracle):
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    buf = np.zeros(256, dtype=np.complex64)
    r = dev.readStream(rx, [buf], 256, timeoutUs=0)                    # nothing captured yet
    assert (r.ret, r.flags) == (0, 0)
    dev.writeSetting("CLOCK_ADVANCE", 100)
    r = dev.readStream(rx, [buf], 256, timeoutUs=0)                    # clamps to what is there, :934-942
    assert (r.ret, r.flags, r.timeNs) == (100, 4, 0)
    dev.writeSetting("CLOCK_ADVANCE", 1000)
    r = dev.readStream(rx, [buf], 256, timeoutUs=0)
    assert (r.ret, r.timeNs) == (256, oracle.ticks_to_time_ns(100, RATE))
    assert_bit_exact(buf, rx_reference(oracle, 4, 356)[100:356], "non-blocking data")


def test_tx_rules_against_oracle(oracle):
    """writeStream placement (:989-1104) on the virtual clock vs the oracle's restatement."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    dev.activateStream(tx)
    buf = (np.arange(256) / 512.0 + 0.25j).astype(np.complex64)
    pos = 0

    def state():
        clk = int(dev.readSetting("CLOCK_NOW"))
        delay = pos - clk                                              # playback: appl - hw
        return 65536 - delay, delay

    # untimed write right after start: no underrun
    avail, delay = state()
    exp = oracle.tx_step(pos, avail, delay, 256, 256, 0, 0, 100000, RATE)
    assert dev.writeStream(tx, [buf], 256).ret == exp.ret == 256
    pos = exp.position
    assert int(dev.readSetting("TX_POSITION")) == pos == 256
    # the application stalls for 1000 samples: untimed write skips whole periods + margin (:1032-1037)
    dev.writeSetting("CLOCK_ADVANCE", 1000)
    avail, delay = state()
    exp = oracle.tx_step(pos, avail, delay, 256, 256, 0, 0, 100000, RATE)
    assert exp.skipped == ((1000 - 256) // 256 + 2) * 256
    SoapySDR.drainLog()
    assert dev.writeStream(tx, [buf], 256).ret == exp.ret
    pos = exp.position
    assert int(dev.readSetting("TX_POSITION")) == pos
    assert "TX buffer underrun. Forwarding TX stream by %d samples" % exp.skipped in SoapySDR.drainLog()
    # timestamp in the past: dropped but reported as written (:1013-1023)
    avail, delay = state()
    exp = oracle.tx_step(pos, avail, delay, 256, 256, 4, 1000, 100000, RATE)
    assert exp.discarded == 1
    assert dev.writeStream(tx, [buf], 256, flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=1000).ret == 256
    assert int(dev.readSetting("TX_POSITION")) == pos
    assert "Discarding TX" in SoapySDR.drainLog()
    # timestamp in the future: forwarded to the exact sample
    t = oracle.ticks_to_time_ns(pos + 5000, RATE)
    avail, delay = state()
    exp = oracle.tx_step(pos, avail, delay, 256, 256, 4, t, 100000, RATE)
    assert dev.writeStream(tx, [buf], 256, flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=t).ret == 256
    pos = exp.position
    assert int(dev.readSetting("TX_POSITION")) == pos == exp.position
    # hardware time = playback position in ns (:1107-1139)
    assert dev.getHardwareTime() == oracle.ticks_to_time_ns(int(dev.readSetting("CLOCK_NOW")), RATE)
    # PTT keying: default threshold 1e-3; the ramp's first sample has |s| = 0.25 -> all keyed
    assert int(dev.readSetting("TX_PTT_SAMPLES")) == 256 * 3


def test_deactivate_resets_positions(oracle):
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    dev.activateStream(tx)
    buf = np.zeros(512, dtype=np.complex64)
    assert dev.readStream(rx, [buf], 512).ret == 512
    assert dev.writeStream(tx, [buf], 512).ret == 512
    dev.deactivateStream(rx)
    assert int(dev.readSetting("RX_POSITION")) == 512                  # only one side stopped
    dev.deactivateStream(tx)                                           # both inactive -> reset, :850-854
    assert int(dev.readSetting("RX_POSITION")) == 0 and int(dev.readSetting("TX_POSITION")) == 0
    dev.activateStream(rx)
    first = buf.copy()
    r = dev.readStream(rx, [buf], 512)
    assert r.timeNs == 0 and np.array_equal(first.view(np.uint64), buf.view(np.uint64))


def test_linked_streams(oracle):
    """SoapySX/test/test_linked_streams.py: link=1, prefill TX, then lock-step read/write."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"link": "1"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"link": "1"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    buf = np.zeros(256, dtype=np.complex64)
    initial = np.zeros(256 * 4, dtype=np.complex64)
    clk0 = int(dev.readSetting("CLOCK_NOW"))
    dev.writeSetting("CLOCK_ADVANCE", 5000)                            # nothing runs before the first write
    assert dev.writeStream(tx, [initial], len(initial)).ret == 1024    # starts both streams
    ref = rx_reference(oracle, 4, 256 * 40)
    for i in range(40):
        r = dev.readStream(rx, [buf], len(buf))
        assert r.ret == 256 and r.timeNs == oracle.ticks_to_time_ns(256 * i, RATE)
        assert_bit_exact(buf, ref[256 * i:256 * (i + 1)], "linked rx %d" % i)
        assert dev.writeStream(tx, [buf], len(buf)).ret == 256
    assert int(dev.readSetting("CLOCK_NOW")) == clk0 + 5000 + 256 * 40
    # TX starves: both linked streams stop (:36-43, :497-501)
    dev.writeSetting("CLOCK_ADVANCE", 2000)
    assert dev.writeStream(tx, [buf], len(buf)).ret == SoapySDR.SOAPY_SDR_UNDERFLOW
    assert dev.readStream(rx, [buf], len(buf)).ret == SoapySDR.SOAPY_SDR_OVERFLOW


def test_decim8_interp8_chain(oracle):
    """BASELINE config 3 shape: 256-tap decimate-by-8 RX and interpolate-by-8 TX, full duplex."""
    dev = make(decim="8", interp="8")
    assert dev.readSetting("RX_NTAPS") == "256"
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    dev.activateStream(tx)
    ref = rx_reference(oracle, 8, 1024 * 4)
    buf = np.zeros(1024, dtype=np.complex64)
    latency = 2048                                                     # > one block, or the write is in the past
    dt = oracle.ticks_to_time_ns(latency, RATE)
    stream = np.zeros(latency + 4096, dtype=np.complex64)
    for i in range(4):
        r = dev.readStream(rx, [buf], 1024)
        assert r.ret == 1024
        assert_bit_exact(buf, ref[1024 * i:1024 * (i + 1)], "rx d8 block %d" % i)
        assert dev.writeStream(tx, [buf], 1024, flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=r.timeNs + dt).ret == 1024
        assert int(dev.readSetting("TX_POSITION")) == 1024 * (i + 1) + latency
        stream[latency + 1024 * i: latency + 1024 * (i + 1)] = buf
    got = dev.txCapture(0, len(stream) * 8)
    assert_bit_exact(got, tx_reference(oracle, 8, stream), "dac stream L=8")


def test_wall_clock_mode():
    """Free-running clock: a blocking read of one period takes about period / rate."""
    import time
    dev = SoapySDR.Device({"driver": "sx", "clock": "wall"})
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 600000.0)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "8192"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    assert dev.getHardwareTime() == 0                                  # hardware time is the TX side's clock
    dev.activateStream(rx)                                             # linked: both PCMs start together
    dev.activateStream(tx)
    buf = np.zeros(60000, dtype=np.complex64)
    t0 = time.time()
    r = dev.readStream(rx, [buf], len(buf))
    dt = time.time() - t0
    assert r.ret == 60000 and r.timeNs == 0
    assert 0.08 <= dt < 1.0                                            # 0.1 s of signal
    t1 = dev.getHardwareTime()
    time.sleep(0.05)
    assert dev.getHardwareTime() - t1 >= 40_000_000


def test_control_surface_register_shadow(oracle):
    """SoapySX/test/test.py and test_gains.py as assertions: tuning word, gain split, antennas,
    raw register access on the SX1255 register shadow (SoapySX.cpp:1225-1561)."""
    dev = SoapySDR.Device({"driver": "sx", "clock": "virtual"})         # as constructed: no setSampleRate yet
    RX, TX = SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_TX
    # start-up state: init_registers + RX/TX enabled, both synthesizers at 433.92 MHz, I2S dividers for master clock / 256
    regs = dev.readRegisters("", 0, 0x14)
    assert regs[0] == 0x0F and regs[7] == 0x11 and regs[0x11] == 3 and regs[0x12:0x14] == [0x22, 0x2C]
    # ... which setSampleRate reprograms (:1197-1203; 75 kS/s = divider 512: clkout 3, m 0, n 6), leaving the rest alone
    dev.setSampleRate(RX, 0, RATE)
    after = dev.readRegisters("", 0, 0x14)
    assert after[0x12:0x14] == [0x23, 0x34] and after[:0x12] == regs[:0x12]
    f0, w0 = oracle.quantize_frequency(38.4e6, 433.92e6)
    assert dev.getFrequency(RX, 0) == f0 == dev.getFrequency(TX, 0)
    assert (regs[1] << 16 | regs[2] << 8 | regs[3]) == w0
    for f in (432.55e6, 434.55e6, 0.0, 1e9):
        dev.setFrequency(RX, 0, f)
        assert dev.getFrequency(RX, 0) == oracle.quantize_frequency(38.4e6, f)[0]
    assert dev.getFrequency(TX, 0) == f0                                   # TX word untouched
    # ranges test_gains.py prints: overall = element ranges added up (SoapySDR's default), elements :1291-1306
    r = dev.getGainRange(RX, 0)
    assert (r.minimum(), r.maximum()) == (0.0, 78.0)
    r = dev.getGainRange(TX, 0)
    assert (r.minimum(), r.maximum()) == (0.0, 39.0)
    r = dev.getGainRange(RX, 0, "LNA")
    assert (r.minimum(), r.maximum(), r.step()) == (0.0, 48.0, 6.0)
    r = dev.getGainRange(TX, 0, "MIXER")
    assert (r.minimum(), r.maximum(), r.step()) == (0.0, 30.0, 2.0)
    # gain sweep of test_gains.py
    assert dev.listGains(RX, 0) == ["LNA", "PGA"] and dev.listGains(TX, 0) == ["DAC", "MIXER"]
    for g in range(-10, 90):
        dev.setGain(RX, 0, float(g))
        lna, pga = oracle.gain_split(1, g)
        assert (dev.getGain(RX, 0, "LNA"), dev.getGain(RX, 0, "PGA")) == (lna, pga), g
        assert dev.getGain(RX, 0) == lna + pga
    for g in range(-10, 50):
        dev.setGain(TX, 0, float(g))
        assert (dev.getGain(TX, 0, "DAC"), dev.getGain(TX, 0, "MIXER")) == oracle.gain_split(0, g), g
    dev.setGain(RX, 0, "PGA", 7.0)
    assert dev.getGain(RX, 0, "PGA") == 8.0                                # round(3.5) = 4 steps of 2 dB
    # antennas
    assert dev.listAntennas(RX, 0) == ["RX", "LB"] and dev.listAntennas(TX, 0) == ["TX", "NONE"]
    assert dev.getAntenna(RX, 0) == "RX" and dev.getAntenna(TX, 0) == "TX"
    dev.setAntenna(RX, 0, "LB")
    dev.setAntenna(TX, 0, "NONE")
    assert dev.getAntenna(RX, 0) == "LB" and dev.getAntenna(TX, 0) == "NONE"
    assert dev.readRegister("", 0x10) & 0x0C == 0x04 and dev.readRegister("", 0x00) & 0x08 == 0
    # raw registers: test.py writes one register, then expects an exception for 3 registers at 0x7E
    dev.writeRegister("", 0x0D, 0x2B)
    assert dev.readRegister("", 0x0D) == 0x2B
    with pytest.raises(RuntimeError, match="Invalid register address"):
        dev.writeRegisters("", 0x7E, [1, 2, 3])
    dev.writeSetting("PA", "AUTO")


def test_s32_wire_mode(oracle):
    """Device arg wire=s32: the synthetic chip side speaks the reference's S32_LE I2S words; what the
    application sees through readStream is unchanged (the source words convert back exactly) and the
    synthetic DAC holds convert_tx_buffer words with the keying bits."""
    dev = make(wire="s32")
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0.05"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    ref = rx_reference(oracle, 4, 2048)
    buf = np.zeros(1024, dtype=np.complex64)
    sent = np.zeros(2048, dtype=np.complex64)
    for i in range(2):
        assert dev.readStream(rx, [buf], 1024).ret == 1024
        assert_bit_exact(buf, ref[1024 * i:1024 * (i + 1)], "s32 wire rx block %d" % i)
        assert dev.writeStream(tx, [buf], 1024).ret == 1024
        sent[1024 * i:1024 * (i + 1)] = buf
    L = int(dev.readSetting("TX_INTERP"))
    start = int(dev.readSetting("TX_POSITION")) - 2048          # first write began after the startup underrun skip
    stream = np.zeros(start + 2048, dtype=np.complex64)
    stream[start:] = sent
    words = dev.txCapture(0, len(stream) * L).view(np.int32)
    thr2 = np.float32(0.05) * np.float32(0.05)
    assert np.array_equal(words, oracle.convert_tx(tx_reference(oracle, L, stream), thr2))
    keyed = (words[0::2] & 3) == 3
    assert keyed.any() and not keyed[: (start - 40) * L].any()


def test_rx_and_tx_threads(oracle):
    """example/plot_rxtx_response.py:65-77 runs TX in its own thread next to the RX loop: the per-stream
    mutexes (SoapySX.cpp:373, :878, :979) must let both sides stream concurrently."""
    import threading
    dev = SoapySDR.Device({"driver": "sx", "clock": "wall"})
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 600000.0)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "4096"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "4096", "threshold": "0"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    nblk, blk = 40, 4096
    errors = []

    def tx_loop():
        buf = np.full(blk, 0.5 + 0.25j, dtype=np.complex64)
        for _ in range(nblk):
            r = dev.writeStream(tx, [buf], blk)
            if r.ret != blk:
                errors.append(("tx", r.ret))
            dev.getHardwareTime()

    t = threading.Thread(target=tx_loop)
    t.start()
    got = np.zeros(nblk * blk, dtype=np.complex64)
    buf = np.zeros(blk, dtype=np.complex64)
    times = []
    for i in range(nblk):
        r = dev.readStream(rx, [buf], blk)
        if r.ret != blk:
            errors.append(("rx", r.ret))
        times.append(r.timeNs)
        got[i * blk:(i + 1) * blk] = buf
    t.join()
    assert not errors, errors
    # no overrun at these rates: contiguous positions, data equals the stream from position 0
    assert times == [oracle.ticks_to_time_ns(i * blk, 600000.0) for i in range(nblk)]
    h = sxxcvr_amd.design_lowpass(128, 4)
    ref = oracle.decim_f32(h, 4, oracle.synth_iq(SEED, 0, 0, 4 * nblk * blk), 2, 4)
    assert_bit_exact(got, ref, "threaded rx stream")
    assert int(dev.readSetting("TX_WRITTEN")) == nblk * blk


def test_rx_and_tx_threads_with_megabyte_blocks(oracle):
    """The same two threads with blocks of 2^18 samples (2 MiB) at 600 kS/s on the wall clock (0.44 s a block), ordinary
    and page-locked buffers alternating: both chains run their DMA-copy paths, copy pools and second streams at the
    same time.  The RX stream must be the oracle's, contiguous from position 0; the TX side must have taken every
    sample and counted the keyed ones."""
    import threading
    dev = SoapySDR.Device({"driver": "sx", "clock": "wall"})
    rate = 600000.0
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, rate)
    dev.setSampleRate(SoapySDR.SOAPY_SDR_TX, 0, rate)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536", "threshold": "0.5"})
    nblk, blk = 6, 1 << 18
    rng = np.random.default_rng(5)
    sent = (rng.uniform(-1, 1, blk) + 1j * rng.uniform(-1, 1, blk)).astype(np.complex64)
    f = sent.view(np.float32).reshape(-1, 2)
    keyed_per_block = int(np.count_nonzero(f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1] >= np.float32(0.25)))
    pin_rx = sxxcvr_amd.pin_array(np.zeros(blk, dtype=np.complex64))
    pin_tx = sxxcvr_amd.pin_array(sent.copy())
    errors = []
    try:
        dev.activateStream(rx)
        dev.activateStream(tx)

        def tx_loop():
            for i in range(nblk):
                r = dev.writeStream(tx, [pin_tx if i % 2 else sent], blk)
                if r.ret != blk:
                    errors.append(("tx", i, r.ret))

        t = threading.Thread(target=tx_loop)
        t.start()
        got = np.zeros(nblk * blk, dtype=np.complex64)
        plain = np.zeros(blk, dtype=np.complex64)
        times = []
        for i in range(nblk):
            buf = pin_rx if i % 2 else plain
            r = dev.readStream(rx, [buf], blk)
            if r.ret != blk:
                errors.append(("rx", i, r.ret))
            times.append(r.timeNs)
            got[i * blk:(i + 1) * blk] = buf
        t.join()
        assert not errors, errors
        assert times == [oracle.ticks_to_time_ns(i * blk, rate) for i in range(nblk)]
        h = sxxcvr_amd.design_lowpass(128, 4)
        ref = oracle.decim_f32(h, 4, oracle.synth_iq_mt(SEED, 0, 0, 4 * nblk * blk, 8), 2, 4, threads=8)
        assert_bit_exact(got, ref, "threaded rx stream, megabyte blocks")
        assert int(dev.readSetting("TX_WRITTEN")) == nblk * blk
        assert int(dev.readSetting("TX_PTT_SAMPLES")) == nblk * keyed_per_block
    finally:
        sxxcvr_amd.unpin_array(pin_rx)
        sxxcvr_amd.unpin_array(pin_tx)


def test_multi_channel_device(oracle):
    """Device argument channels=N (the reference has one channel, SX.cpp:1591-1595; BASELINE config 4 puts
    8 on a GPU): one stream carries all N channels, buffs[c] = channel c of the synthetic source, and the TX
    sink keeps one ring per channel.  Every channel is checked against the oracle bit for bit."""
    n = 4
    dev = make(channels=str(n), first_channel="8", decim="4", interp="4")
    assert dev.getNumChannels(SoapySDR.SOAPY_SDR_RX) == n and dev.getNumChannels(SoapySDR.SOAPY_SDR_TX) == n
    with pytest.raises(RuntimeError, match="all channels"):
        dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, list(range(n)), {})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [], {"threshold": "0"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    h = sxxcvr_amd.design_lowpass(128, 4)
    total, got = 0, [[] for _ in range(n)]
    for block in (256, 256, 1000, 5000, 256):                    # crosses read-ahead batch boundaries (4096, ...)
        bufs = [np.zeros(block, dtype=np.complex64) for _ in range(n)]
        r = dev.readStream(rx, bufs, block)
        assert r.ret == block and r.timeNs == oracle.ticks_to_time_ns(total, RATE)
        for c in range(n):
            got[c].append(bufs[c].copy())
        total += block
    for c in range(n):
        ref = oracle.decim_f32(h, 4, oracle.synth_iq(SEED, 8 + c, 0, 4 * total), 2, 4)
        assert_bit_exact(np.concatenate(got[c]), ref, "rx channel %d" % c)
    # TX: a different block per channel, written at the stream position the device is at
    ht = sxxcvr_amd.design_lowpass(128, 4, 8.0, 4.0)
    blocks = [oracle.synth_iq(77, c, 0, 2048) * np.float32(0.5) for c in range(n)]
    pos0 = int(dev.readSetting("TX_POSITION"))
    w = dev.writeStream(tx, blocks, 2048)
    assert w.ret == 2048
    first = int(dev.readSetting("TX_POSITION")) - 2048
    assert first >= pos0
    for c in range(n):
        dev.writeSetting("TX_CAPTURE_CHANNEL", str(c))
        out = dev.txCapture(first * 4, 2048 * 4)
        stream = np.concatenate([np.zeros(32, dtype=np.complex64), blocks[c]])
        assert_bit_exact(out, oracle.interp_f32(ht, 4, stream, 2)[32 * 4:], "tx channel %d" % c)
    with pytest.raises(RuntimeError, match="No such channel"):
        dev.writeSetting("TX_CAPTURE_CHANNEL", str(n))


def test_rx_read_ahead_is_invisible(oracle):
    """The RX chain produces the stream in batches and keeps the next batch in flight; whatever the read
    sizes, and across an overrun skip (a jump of the position drops the batches and re-primes the filter),
    the samples are those of one pass of the oracle over the same stream positions."""
    dev = make(decim="8")
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "256"})
    dev.activateStream(rx)
    ref = rx_reference(oracle, 8, 1 << 18)
    pos = 0
    rng = np.random.default_rng(5)
    for n in [1, 255, 256, 4095, 4097, 70000] + [int(v) for v in rng.integers(1, 9000, size=25)]:
        buf = np.zeros(n, dtype=np.complex64)
        r = dev.readStream(rx, [buf], n)
        assert r.ret == n and r.timeNs == oracle.ticks_to_time_ns(pos, RATE)
        assert_bit_exact(buf, ref[pos:pos + n], "read of %d at %d" % (n, pos))
        pos += n
    # let the ring overflow: the next read skips ahead (SX.cpp:910-927) and the data follows the position
    dev.writeSetting("CLOCK_ADVANCE", str(65536 + 3 * 256 + 17))
    buf = np.zeros(300, dtype=np.complex64)
    r = dev.readStream(rx, [buf], 300)
    new_pos = int(dev.readSetting("RX_POSITION")) - 300
    assert new_pos > pos and (new_pos - pos) % 256 == 0
    assert r.timeNs == oracle.ticks_to_time_ns(new_pos, RATE)
    assert_bit_exact(buf, ref[new_pos:new_pos + 300], "after the skip")


def test_ratio_follows_sample_rate(oracle):
    """decim=auto / interp=auto: like the SX1255, whose decimator and interpolator follow the divider that
    setSampleRate programs (SX.cpp:1192-1208), the ratio is divider / 16.  The reference's own test rates give
    BASELINE's shapes: 300 kS/s -> 256-tap /8 (config 3), 75 kS/s -> 1024-tap /32 (config 5)."""
    dev = SoapySDR.Device({"driver": "sx", "clock": "virtual", "decim": "auto", "interp": "auto"})
    assert dev.readSetting("RX_DECIM") == "16"                      # default rate = master clock / 256
    for rate, ratio in ((300000.0, 8), (75000.0, 32)):
        dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, rate)
        assert dev.readSetting("RX_DECIM") == str(ratio) and dev.readSetting("TX_INTERP") == str(ratio)
        assert dev.readSetting("RX_NTAPS") == str(32 * ratio)
        rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
        tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0"})
        dev.activateStream(rx)
        dev.activateStream(tx)
        ref = rx_reference(oracle, ratio, 3000)
        buf = np.zeros(3000, dtype=np.complex64)
        r = dev.readStream(rx, [buf], 3000)
        assert r.ret == 3000 and r.timeNs == 0
        assert_bit_exact(buf, ref, "rx at %g S/s" % rate)
        block = oracle.synth_iq(5, 1, 0, 1024) * np.float32(0.25)
        w = dev.writeStream(tx, [block], 1024)
        assert w.ret == 1024
        first = int(dev.readSetting("TX_POSITION")) - 1024
        out = dev.txCapture(first * ratio, 1024 * ratio)
        stream = np.concatenate([np.zeros(32, dtype=np.complex64), block])
        assert_bit_exact(out, tx_reference(oracle, ratio, stream)[32 * ratio:], "tx at %g S/s" % rate)
        dev.deactivateStream(rx)
        dev.deactivateStream(tx)
        dev.closeStream(rx)
        dev.closeStream(tx)
    # 50 kS/s = master clock / 768 -> ratio 48: twelve column groups meet in the adjacent-pair tree with an odd level, under
    # the rotated contract (sxfir_contract_rotation = 1): decim_blocks_kernel, or the generic kernel for ragged calls
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 50000.0)
    assert dev.readSetting("RX_DECIM") == "48" and dev.readSetting("RX_NTAPS") == str(32 * 48)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    buf = np.zeros(700, dtype=np.complex64)
    assert dev.readStream(rx, [buf], 700).ret == 700
    h = sxxcvr_amd.design_lowpass(32 * 48, 48)
    assert_bit_exact(buf, oracle.decim_f32(h, 48, oracle.synth_iq(SEED, 0, 0, 700 * 48), 2, 4, rot=1), "rx at 50 kS/s")
    dev.deactivateStream(rx)
    dev.closeStream(rx)
    # ... and the TX side of the two slowest rates: x48 / x96 as phase blocks of the tile kernels
    for rate, ratio in ((50000.0, 48), (25000.0, 96)):
        dev.setSampleRate(SoapySDR.SOAPY_SDR_TX, 0, rate)
        assert dev.readSetting("TX_INTERP") == str(ratio)
        tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0"})
        dev.activateStream(tx)
        block = oracle.synth_iq(5, 2, 0, 1500) * np.float32(0.25)
        assert dev.writeStream(tx, [block], 1500).ret == 1500
        first = int(dev.readSetting("TX_POSITION")) - 1500
        out = dev.txCapture(first * ratio, 1500 * ratio)
        stream = np.concatenate([np.zeros(32, dtype=np.complex64), block])
        assert_bit_exact(out, tx_reference(oracle, ratio, stream)[32 * ratio:], "tx at %g S/s" % rate)
        dev.deactivateStream(tx)
        dev.closeStream(tx)
    # 25 kS/s = master clock / 1536 -> ratio 96, 3072 taps: the carried-over history (3072 samples) is longer
    # than one workgroup used to hold; read enough for several RX batches so that the history is carried
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 25000.0)
    assert dev.readSetting("RX_DECIM") == "96" and dev.readSetting("RX_NTAPS") == str(32 * 96)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    dev.activateStream(rx)
    n_out = 3 * 4096 + 500
    got = np.zeros(n_out, dtype=np.complex64)
    pos = 0
    while pos < n_out:
        m = min(1000, n_out - pos)
        buf = np.zeros(m, dtype=np.complex64)
        assert dev.readStream(rx, [buf], m).ret == m
        got[pos:pos + m] = buf
        pos += m
    h = sxxcvr_amd.design_lowpass(32 * 96, 96)
    assert_bit_exact(got, oracle.decim_f32(h, 96, oracle.synth_iq(SEED, 0, 0, n_out * 96), 2, 4, rot=1), "rx at 25 kS/s")
    dev.deactivateStream(rx)
    dev.closeStream(rx)
    with pytest.raises(RuntimeError, match="Unsupported sample rate"):
        dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 48000.0)


def test_repeater_example_runs():
    """examples/repeater_loopback.py: the reference's linear-repeater call pattern for a fixed number of blocks."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--rate", "300000", "--blocks", "60"]):
        run = subprocess.run([sys.executable, os.path.join(root, "examples", "repeater_loopback.py")] + extra,
                             capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
        assert "TX placed at sample" in run.stdout


def test_large_reads_into_registered_memory_skip_the_staging_copy(oracle):
    """readStream with a block of >= 32768 samples into page-locked caller memory (sxxcvr_amd.pin_array = 
    sxfir_host_register): the decimator stores straight into the caller's buffer.  Same samples as the staged
    path, which ordinary (pageable) memory of the same size takes; small reads continue the stream seamlessly."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536"})
    dev.activateStream(rx)
    n = 1 << 17
    ref = rx_reference(oracle, 4, 3 * n + 1000)
    small = np.zeros(1000, dtype=np.complex64)
    assert dev.readStream(rx, [small], 1000).ret == 1000
    assert_bit_exact(small, ref[:1000], "first small read")
    pinned = sxxcvr_amd.pin_array(np.zeros(n, dtype=np.complex64))
    try:
        pos = 1000
        for rep in range(2):
            pinned[:] = 0
            assert dev.readStream(rx, [pinned], n).ret == n
            assert_bit_exact(pinned, ref[pos:pos + n], "registered buffer, read %d" % rep)
            pos += n
        direct = int(dev.readSetting("RX_DIRECT_SAMPLES"))
        assert direct >= n                                   # at least the first large read went direct
        plain = np.zeros(n, dtype=np.complex64)
        assert dev.readStream(rx, [plain], n).ret == n
        assert_bit_exact(plain, ref[pos:pos + n], "pageable buffer")
        assert int(dev.readSetting("RX_DIRECT_SAMPLES")) == direct
    finally:
        sxxcvr_amd.unpin_array(pinned)
    dev.deactivateStream(rx)
    dev.closeStream(rx)


def test_megabyte_blocks_cross_pcie_as_dma_copies_beside_the_kernels(oracle):
    """Reads and writes of 2^19 samples (4 MiB): the decimated block lands in HBM and a DMA-engine copy on a second
    stream takes it to the host while the next pass runs (RX: staged batches and page-locked caller memory alike);
    a written block is copied into grown pinned slots by the copy pool, or taken from page-locked memory as it
    is, and reaches HBM by DMA copies before the interpolator and the keying count read it.  Every sample is
    compared with the oracle, on both sides, with small calls in between."""
    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536", "threshold": "0.5"})
    dev.activateStream(rx)
    n = 1 << 19
    h = sxxcvr_amd.design_lowpass(128, 4)
    total = 4 * n + 1300
    ref = oracle.decim_f32(h, 4, oracle.synth_iq_mt(SEED, 0, 0, 4 * total, 8), 2, 4, threads=8)
    pinned = sxxcvr_amd.pin_array(np.zeros(n, dtype=np.complex64))
    try:
        pos = 0
        for what, m in (("small", 1000), ("registered", n), ("pageable", n), ("pageable", n), ("small", 300), ("registered", n)):
            buf = pinned if what == "registered" else np.zeros(m, dtype=np.complex64)
            buf[:] = 0
            assert dev.readStream(rx, [buf], m).ret == m
            assert_bit_exact(buf[:m], ref[pos:pos + m], "%s read of %d at %d" % (what, m, pos))
            pos += m
        # the first registered read reached the caller's memory without a host copy (bar what the small read before
        # it had left in the pinned staging slots: at most two batches of 4096)
        assert int(dev.readSetting("RX_DIRECT_SAMPLES")) >= n - 8192
        dev.deactivateStream(rx)

        dev.activateStream(tx)
        rng = np.random.default_rng(11)
        stream = np.zeros(4 * n, dtype=np.complex64)
        keyed = 0
        end = 0
        for what, m in (("small", 700), ("pageable", n), ("registered", n), ("small", 256), ("pageable", n // 2)):
            x = (rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(np.complex64)
            f = x.view(np.float32).reshape(-1, 2)
            keyed += int(np.count_nonzero(f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1] >= np.float32(0.25)))
            src = x
            if what == "registered":
                pinned[:m] = x
                src = pinned
            assert dev.writeStream(tx, [src], m).ret == m
            if what == "registered":
                pinned[:m] = 0                               # the call has returned: the memory is the caller's again
            end = int(dev.readSetting("TX_POSITION"))
            stream[end - m:end] = x
        assert int(dev.readSetting("TX_DIRECT_SAMPLES")) == n
        assert int(dev.readSetting("TX_PTT_SAMPLES")) == keyed and keyed > 0
        L = int(dev.readSetting("TX_INTERP"))
        want = tx_reference(oracle, L, stream[:end])
        tail = 1 << 19                                        # stream samples of the sink's tail to compare
        got = dev.txCapture((end - tail) * L, tail * L)
        assert_bit_exact(got, want[(end - tail) * L:], "dac stream tail")
    finally:
        sxxcvr_amd.unpin_array(pinned)


def test_random_call_sequences_keep_the_stream_intact(oracle):
    """Forty reads and thirty writes of random sizes (one sample … 4 MiB), into ordinary and page-locked buffers in
    any order: every change of path inside the chains (staged / in HBM / direct, zero-copy / DMA copy, slot growth)
    must hand over the same stream the oracle computes."""
    rng = np.random.default_rng(2026)

    def block():
        kind = rng.integers(0, 6)
        return int([rng.integers(1, 300), 256 * rng.integers(1, 17), rng.integers(30000, 40000),
                    rng.integers(131072, 200000), rng.integers(262144, 524288), rng.integers(1, 5000)][kind])

    dev = make()
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": "65536", "threshold": "0.5"})
    pinned = sxxcvr_amd.pin_array(np.zeros(1 << 19, dtype=np.complex64))
    try:
        sizes = [block() for _ in range(40)]
        total = sum(sizes)
        h = sxxcvr_amd.design_lowpass(128, 4)
        ref = oracle.decim_f32(h, 4, oracle.synth_iq_mt(SEED, 0, 0, 4 * total, 8), 2, 4, threads=8)
        dev.activateStream(rx)
        pos = 0
        for i, m in enumerate(sizes):
            locked = bool(rng.integers(0, 2))
            buf = pinned if locked else np.zeros(m, dtype=np.complex64)
            buf[:m] = 0
            assert dev.readStream(rx, [buf], m).ret == m
            assert_bit_exact(buf[:m], ref[pos:pos + m], "read %d: %d samples at %d, %s" % (i, m, pos, "page-locked" if locked else "pageable"))
            pos += m
        dev.deactivateStream(rx)

        dev.activateStream(tx)
        sizes = [block() for _ in range(30)]
        stream = np.zeros(sum(sizes) + 70000 * 30, dtype=np.complex64)
        keyed, end = 0, 0
        for i, m in enumerate(sizes):
            x = (rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(np.complex64)
            f = x.view(np.float32).reshape(-1, 2)
            keyed += int(np.count_nonzero(f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1] >= np.float32(0.25)))
            locked = bool(rng.integers(0, 2))
            src = x
            if locked:
                pinned[:m] = x
                src = pinned
            assert dev.writeStream(tx, [src], m).ret == m
            if locked:
                pinned[:m] = 0
            end = int(dev.readSetting("TX_POSITION"))
            stream[end - m:end] = x
        assert int(dev.readSetting("TX_PTT_SAMPLES")) == keyed
        L = int(dev.readSetting("TX_INTERP"))
        tail = min(end, 1 << 20)
        want = tx_reference(oracle, L, stream[:end])
        got = dev.txCapture((end - tail) * L, tail * L)
        assert_bit_exact(got, want[(end - tail) * L:], "dac stream tail")
    finally:
        sxxcvr_amd.unpin_array(pinned)


def test_keying_count_on_the_gpu_matches_the_reference_rule(oracle):
    """TX_PTT_SAMPLES: the number of written samples whose squared magnitude reaches threshold^2 (the PTT bit of
    convert_tx_buffer, SX.cpp:126-135), counted by the GPU as the staged blocks pass; silence from timed gaps
    is not counted."""
    dev = make()
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0.5"})
    dev.activateStream(tx)
    rng = np.random.default_rng(4)
    want = 0
    for blk in (256, 1000, 5000, 256):
        x = (rng.uniform(-1, 1, blk) + 1j * rng.uniform(-1, 1, blk)).astype(np.complex64)
        f = x.view(np.float32).reshape(-1, 2)
        want += int(np.count_nonzero(f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1] >= np.float32(0.25)))
        assert dev.writeStream(tx, [x], blk).ret == blk
    assert int(dev.readSetting("TX_PTT_SAMPLES")) == want and want > 0
    dev.deactivateStream(tx)
    dev.closeStream(tx)


@pytest.mark.parametrize("shards_per_gpu", [1, 2])
def test_config4_channel_shards_through_the_device_path(oracle, shards_per_gpu):
    """BASELINE config 4 in its natural SoapySDR form: one process, one Device per shard of 8 channels (gpu=k,
    channels=8, first_channel=8k; one Device per visible GPU, times shards_per_gpu to exercise several shards on a
    1-GPU box), one reader thread each, no inter-GPU traffic (buffs[c] are host buffers).  Every channel of every
    shard is compared with the oracle; all shards run concurrently."""
    import threading
    import torch
    ngpu = torch.cuda.device_count()
    shards = [(g, s) for g in range(ngpu) for s in range(shards_per_gpu)]
    per, n_read, blocks = 8, 20000, (256, 4096, 256, 15000, 392)
    assert sum(blocks) == n_read
    devs = []
    for k, (g, s) in enumerate(shards):
        devs.append(make(gpu=str(g), channels=str(per), first_channel=str(per * k)))
    results, errors = [None] * len(shards), []

    def reader(k):
        try:
            dev = devs[k]
            rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, list(range(per)), {})
            dev.activateStream(rx)
            got = [[] for _ in range(per)]
            for b in blocks:
                bufs = [np.zeros(b, dtype=np.complex64) for _ in range(per)]
                r = dev.readStream(rx, bufs, b)
                assert r.ret == b
                for c in range(per):
                    got[c].append(bufs[c])
            dev.deactivateStream(rx)
            dev.closeStream(rx)
            results[k] = [np.concatenate(g) for g in got]
        except Exception as e:                               # surfaced in the main thread below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=reader, args=(k,)) for k in range(len(shards))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    h = sxxcvr_amd.design_lowpass(128, 4)
    for k in range(len(shards)):
        for c in range(per):
            ref = oracle.decim_f32(h, 4, oracle.synth_iq(SEED, per * k + c, 0, 4 * n_read), 2, 4)
            assert_bit_exact(results[k][c], ref, "shard %d (gpu %d) channel %d" % (k, shards[k][0], per * k + c))


def test_kernel_stores_into_host_memory_are_visible_after_the_wait(tmp_path):
    """The mechanism the RX chain's small passes rely on (GpuChains.hpp: the decimator stores straight into the chain's
    hipHostMalloc'ed staging, an event, then the host reads): tools/hostvis_probe.hip writes a per-launch pattern in the
    decimator's store shape (whole lines + a ragged tile ending inside a line), waits, and checks EVERY word -- here
    20000 launches for each of the 12 cells on hipHostMalloc'ed memory (nt / plain stores x three kinds of wait x idle /
    beside a streaming kernel); profiles/round4c_hostvis_probe.txt holds the 10^6-launch runs of all the cells."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "hostvis_probe")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-w", root + "/tools/hostvis_probe.hip", "-o", exe])
    run = subprocess.run([exe, "20000", "hipHostMalloc default"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:]
    lines = [l for l in run.stdout.splitlines() if "launches with stale words" in l and not l.startswith("#")]
    assert len(lines) == 12 and all(" : 0 of 20000 " in l for l in lines), run.stdout[-2000:]
