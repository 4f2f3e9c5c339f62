"""Host logic behind the Device on the CPU (no GPU): the sample-clock / PCM model that replaces ALSA and the
SX1255 register shadow, driven by a small C++ probe (tests/host/host_logic_probe.cpp, built with g++ here)
and compared with hand-evaluated ALSA semantics (SoapySX.cpp:434-517 sets exactly these modes) and with the
oracle's restatement of the reference's gain / tuning arithmetic."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "host_logic_probe")
    csrc = os.path.join(ROOT, "sxxcvr_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           "-I" + os.path.join(csrc, "compat"), os.path.join(ROOT, "tests", "host", "host_logic_probe.cpp"),
           os.path.join(csrc, "compat", "SoapySDRCompat.cpp"), "-o", exe, "-pthread"]
    subprocess.run(cmd, check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        key, *vals = line.split()
        rows.setdefault(key, []).append(vals)
    return rows


def one(rows, key):
    assert len(rows[key]) == 1
    return rows[key][0]


def test_ring_geometry(probe):
    """AlsaPcm::configure (SX.cpp:451-466): period defaults to 256, caps at 65536; the ring is the largest
    multiple of the period within 65536 frames."""
    want = {0: (256, 65536), 256: (256, 65536), 1000: (1000, 65000), 65536: (65536, 65536), 100000: (65536, 65536)}
    for period, got_p, got_b in probe["geometry"]:
        assert (int(got_p), int(got_b)) == want[int(period)]


def test_capture_semantics(probe):
    # running capture PCM: avail = delay = frames the clock produced and nobody read
    assert one(probe, "capture_after_1000") == ["0", "1000", "1000"]
    # a read of what is there returns at once and leaves the clock alone
    assert one(probe, "capture_read") == ["256", "0", "1000"]
    # a blocking read waits until the last frame exists: appl 256 + 2000 = clock 2256
    assert one(probe, "capture_blocking_read") == ["2000", "256", "2256"]
    # NORMAL mode has stop_threshold = boundary (SX.cpp:492-496): it keeps running through an overrun
    assert one(probe, "capture_overrun_normal") == ["0", "70000", "1"]
    # snd_pcm_forward moves the application pointer only
    assert one(probe, "capture_forward") == ["5000", "7256"]


def test_link_mode_semantics(probe):
    """LINK mode (SX.cpp:36-43, :497-501): linked PCMs start on the first TX write and stop together on xrun."""
    # prepared playback ring takes at most one ring; the write starts both PCMs
    assert one(probe, "link_first_write") == ["65536", "0", "1", "1"]
    # playback: delay = written - played, avail = ring - delay;  capture: avail = delay = produced
    assert one(probe, "link_tx_after_1000") == ["1000", "64536"]
    assert one(probe, "link_rx_after_1000") == ["1000", "1000"]
    # the ring runs dry: -EPIPE (-32) on both, both stopped
    assert one(probe, "link_underrun") == ["-32", "1", "1"]
    assert one(probe, "link_rx_after_xrun") == ["-32"]
    # drop + prepare + reset (SX.cpp:419-432)
    assert one(probe, "link_after_reset") == ["1", "0", "0"]


def test_register_shadow_against_oracle(probe, oracle):
    """Tuning words and the LNA/PGA and DAC/MIXER splits are the reference's arithmetic (SX.cpp:1236-1272,
    :1370-1394) as restated by the oracle."""
    for clock, f, tuned, word in probe["tune"]:
        want_f, want_w = oracle.quantize_frequency(float(clock), float(f))
        assert int(word) == want_w and abs(float(tuned) - want_f) < 1e-3
    for key, direction in (("rxgain", 1), ("txgain", 0)):          # SOAPY_SDR_RX = 1, SOAPY_SDR_TX = 0
        for g, coarse, fine, _reg in probe[key]:
            want = oracle.gain_split(direction, float(g))
            assert (float(coarse), float(fine)) == want, (key, g)
    # register image after power-up: RX, TX and PA driver enabled in reg 0, default tuning word 433.92 MHz
    boot = [int(v) for v in one(probe, "boot")]
    assert boot[0] == 0x0F and boot[0x11] == 3                      # status: both PLLs locked (SX.cpp:635-636)
    # every other register of the power-up image is the REFERENCE'S OWN init_registers[] (SoapySX.cpp:139-176, compiled from
    # /root/reference into tests/golden/rate_table.json by `make -C oracle ref`): tuning words, gains, filters, I2S dividers
    import json
    init = json.load(open(os.path.join(ROOT, "tests", "golden", "rate_table.json")))["init_registers"]
    assert len(boot) == len(init) == 0x14
    assert [boot[i] for i in range(7, 0x14) if i != 0x11] == [init[i] for i in range(7, 0x14) if i != 0x11]
    assert boot[0] == init[0] | (7 << 1)                            # + RX, TX and PA driver enabled (:625)
    # (registers 1-6: the reference re-tunes to 433.92 MHz once the master clock is known, :659-660, as the shadow does; its
    # static image holds that frequency's word for a 32 MHz clock)
    f32, w32 = oracle.quantize_frequency(32.0e6, 433.92e6)
    assert (init[1] << 16) | (init[2] << 8) | init[3] == w32 == (init[4] << 16) | (init[5] << 8) | init[6]
    f, w = oracle.quantize_frequency(38.4e6, 433.92e6)
    assert (boot[1] << 16) | (boot[2] << 8) | boot[3] == w and (boot[4] << 16) | (boot[5] << 8) | boot[6] == w
    ant = one(probe, "antenna")
    assert ant[:2] == ["DLB", "NONE"] and int(ant[2]) & 0x0C == 0x0C and int(ant[3]) & 0x08 == 0
    assert one(probe, "burst_over_end") == ["Invalid", "register", "address"]


def test_tick_conversion_against_oracle(probe, oracle):
    for rate, t, ns, back in probe["ticks"]:
        assert int(ns) == oracle.ticks_to_time_ns(int(t), float(rate))
        assert int(back) == int(t)                                  # round trip is exact at every table rate


# ------------------------------------------------------------------------------------------------------------------
# The host side under sanitizers.  The Device module is multi-threaded C++ (per-stream mutexes as SoapySX.cpp:373, :750,
# :878, :979, :1110-1125; CopyPool workers; two HIP streams per chain; read-ahead slots).  Everything above the C ABI is
# compiled here as it ships, over tests/host/fake_sxfir.cpp -- a TEST-ONLY backend in which every stream is a thread of
# its own that runs its queue late (random delays), so that a wait the host code forgot shows as wrong samples AND as a
# data race -- and run three ways: AddressSanitizer + UBSan, ThreadSanitizer, plain.  All builds and runs go side by side.
# ------------------------------------------------------------------------------------------------------------------
SAN = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
    "tsan": ["-fsanitize=thread"],
    "plain": [],
}


def _oracle_link():
    """The AVX2+FMA build of the oracle where the CPU has both (8 x faster; same bits): the fake backend's arithmetic."""
    cpuinfo = open("/proc/cpuinfo").read()
    return "-lsxoracle_fast" if (" avx2" in cpuinfo and " fma" in cpuinfo) else "-lsxoracle"


@pytest.fixture(scope="module")
def sanitized(oracle, tmp_path_factory):
    """Builds {chains_probe, device_probe, host_logic_probe, stream_rules_probe} x {asan, tsan[, plain]} and runs them,
    everything in parallel; returns {(probe, flavour): (returncode, stdout, stderr)}."""
    out = tmp_path_factory.mktemp("san")
    csrc = os.path.join(ROOT, "sxxcvr_amd", "csrc")
    odir = os.path.join(ROOT, "oracle")
    host = os.path.join(ROOT, "tests", "host")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-I" + os.path.join(csrc, "compat"), "-I" + odir]
    olink = ["-L" + odir, _oracle_link(), "-Wl,-rpath," + odir, "-pthread"]
    fake = os.path.join(host, "fake_sxfir.cpp")
    device_srcs = [os.path.join(host, "device_probe.cpp"), fake, os.path.join(csrc, "SoapySXHip.cpp"),
                   os.path.join(csrc, "sx_device_capi.cpp"), os.path.join(csrc, "compat", "SoapySDRCompat.cpp")]
    calls = os.environ.get("SX_HOST_PROBE_CALLS")          # 100000 = the full-size run recorded in profiles/round5_host_sanitizers.txt
    jobs = {
        ("chains", "asan"): ([os.path.join(host, "chains_probe.cpp"), fake], olink, []),
        ("chains", "tsan"): ([os.path.join(host, "chains_probe.cpp"), fake], olink, []),
        ("device", "plain"): (device_srcs, olink, [calls or "30000", "40", "lenient"]),
        ("device", "asan"): (device_srcs, olink, [calls or "8000", "40", "lenient"]),
        ("device", "tsan"): (device_srcs, olink, [calls or "4000", "40", "lenient"]),
        ("host_logic", "asan"): ([os.path.join(host, "host_logic_probe.cpp"), os.path.join(csrc, "compat", "SoapySDRCompat.cpp")],
                                 ["-pthread"], []),
        ("host_logic", "tsan"): ([os.path.join(host, "host_logic_probe.cpp"), os.path.join(csrc, "compat", "SoapySDRCompat.cpp")],
                                 ["-pthread"], []),
        ("stream_rules", "asan"): ([os.path.join(host, "stream_rules_probe.cpp")], ["-L" + odir, "-lsxoracle", "-Wl,-rpath," + odir], []),
    }
    builds = {}
    for (probe, flav), (srcs, link, _) in jobs.items():
        exe = str(out / ("%s_%s" % (probe, flav)))
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-ffp-contract=off"] + SAN[flav] + inc + srcs + ["-o", exe] + link
        builds[(probe, flav)] = (exe, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    runs = {}
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    for key, (exe, proc) in builds.items():
        log, _ = proc.communicate(timeout=600)
        assert proc.returncode == 0, "%s %s did not build:\n%s" % (key[0], key[1], log[-3000:])
        runs[key] = subprocess.Popen([exe] + jobs[key][2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    results = {}
    for key, proc in runs.items():
        try:
            so, se = proc.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            proc.kill()
            so, se = proc.communicate()
            se += "\nTIMEOUT"
        results[key] = (proc.returncode, so, se)
    return results


def _clean(results, key):
    rc, so, se = results[key]
    assert "Sanitizer" not in se and "runtime error" not in se and "TIMEOUT" not in se, "%s %s:\n%s" % (key[0], key[1], se[-4000:])
    assert rc == 0, "%s %s exited %d:\n%s\n%s" % (key[0], key[1], rc, so[-2000:], se[-2000:])
    return so.splitlines()


@pytest.mark.parametrize("flavour", ["asan", "tsan"])
def test_gpu_chains_over_fake_backend(sanitized, flavour):
    """GpuChains.hpp (RX batching + read-ahead + jump handling + channel layout, TX write-behind + silence +
    sink-ring wrap) on the CPU: linked against tests/host/fake_sxfir.cpp, a TEST-ONLY asynchronous implementation of
    the C ABI in which the oracle does the arithmetic.  Every sample must equal one direct oracle pass over the stream,
    and neither AddressSanitizer + UBSan nor ThreadSanitizer may have anything to say."""
    lines = _clean(sanitized, ("chains", flavour))
    assert lines[-1] == "bad 0", "\n".join(lines[-30:])
    assert all(l.split()[1] == "0" for l in lines if " mismatches " in l)
    assert "tx_backwards refused" in lines
    # ten sequential reads (154k samples per channel) in a handful of batched passes, not one per read
    launches = int([l for l in lines if l.startswith("rx_launches")][0].split()[1])
    assert launches <= 12
    # a large read into page-locked caller memory is stored by the decimator itself (the first of the two reads of
    # 40000 per channel: the second finds its samples already read ahead in staging); the same size into ordinary
    # memory goes through staging
    assert "rx_direct_samples 40000" in lines and "rx_direct_unchanged 1" in lines
    # passes of a megabyte and more (DMA copies beside the kernels), grown TX slots, page-locked memory on both sides
    assert "rx_large_direct_samples 200000" in lines and "tx_large_direct_samples 150000" in lines
    assert any(l.startswith("rx_random 0 mismatches of 90 reads") for l in lines)
    rk = [l for l in lines if l.startswith("tx_random_keyed ")][0].split()
    assert rk[1] == rk[3] and int(rk[1]) > 0
    assert "rx_large_hbm_samples 800000" in lines            # three more reads of 200000, none through a host copy
    assert "tx_large_slot_frames 262144" in lines
    big = [l for l in lines if l.startswith("tx_large_keyed ")][0].split()
    assert big[1] == big[3] and int(big[1]) > 0
    keyed = [l for l in lines if l.startswith("tx_keyed ")][0].split()
    assert keyed[1] == keyed[3] and int(keyed[1]) > 0 and "tx_keyed_after_reset 0" in lines


@pytest.mark.parametrize("flavour", ["plain", "asan", "tsan"])
def test_device_threads_over_fake_backend(sanitized, flavour):
    """tests/host/device_probe.cpp: SoapySXHip.cpp + sx_device_capi.cpp as they ship, driven through include/sx_device.h
    the way applications do -- an RX thread, a TX thread and a third thread on getHardwareTime / settings / registers
    (example/plot_rxtx_response.py:65-77; SoapySX.cpp:878, :979, :1110-1125) with 4096-sample and megabyte blocks, linked
    streams started and stopped across threads, and thousands of mixed calls (timed writes, non-blocking reads, overrun
    skips, resets) -- every RX block the oracle's at the position its timestamp names, every TX sample accounted for."""
    lines = _clean(sanitized, ("device", flavour))
    assert lines[-1] == "bad 0", "\n".join(lines[-30:])
    assert not [l for l in lines if l.startswith("FAIL")]
    t = [l for l in lines if l.startswith("threads blocks 40 bad 0")]
    m = [l for l in lines if l.startswith("megabyte blocks 6 bad 0")]
    k = [l for l in lines if l.startswith("linked rounds 3 xruns 3 bad 0")]
    c = [l for l in lines if l.startswith("calls ")]
    assert t and m and k and c, lines[-12:]
    f = c[0].split()
    assert int(f[3]) > 1000 and int(f[5]) > 1000 and int(f[7]) > 10 and f[f.index("bad") + 1] == "0"     # reads, writes, clock jumps
    # the register shadow after setSampleRate against the REFERENCE'S OWN table (tests/golden/rate_table.json: sample_rates[],
    # SoapySX.cpp:179-208, compiled from /root/reference by `make -C oracle ref`): 0x12 bits 3-0 = clkout, 0x13 = mant / m / n
    import json
    table = {r["div"]: r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "rate_table.json")))["rows"]}
    regs = [l.split() for l in lines if l.startswith("rate_regs ")]
    assert len(regs) == 2 * len(table) == 12
    for _, clock, rate, r12, r13, r0 in regs:
        row = table[int(round(float(clock) / float(rate)))]
        assert int(r12) & 0x0F == row["clkout"], (clock, rate, r12)
        assert (int(r13) >> 7) & 1 == row["mant"] and (int(r13) >> 6) & 1 == row["m"] and (int(r13) >> 3) & 7 == row["n"], (rate, r13)
        assert (int(r0) >> 1) & 3 == 3                         # RX and TX enabled again, :1207


@pytest.mark.parametrize("flavour", ["asan", "tsan"])
def test_host_logic_probe_under_sanitizers(sanitized, flavour):
    _clean(sanitized, ("host_logic", flavour))


def test_stream_rules_probe_under_sanitizers(sanitized):
    lines = _clean(sanitized, ("stream_rules", "asan"))
    assert lines[-1].strip().endswith("cases 400000 mismatches 0")


def test_stream_placement_rules_against_the_oracle(tmp_path):
    """StreamRules.hpp -- the overrun skip, the non-blocking clamp and the playback placement (timestamp / in
    sequence / past an underrun / in the past) that SoapySXHip::readStream and writeStream apply -- swept over
    400 000 random and edge-case counter sets against the oracle's restatement of SoapySX.cpp:897-1100."""
    exe = str(tmp_path / "stream_rules_probe")
    odir = os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "sxxcvr_amd", "csrc"), "-I" + odir,
           os.path.join(ROOT, "tests", "host", "stream_rules_probe.cpp"), "-o", exe, "-L" + odir, "-lsxoracle",
           "-Wl,-rpath," + odir]
    subprocess.run(cmd, check=True)
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and run.stdout.strip().endswith("cases 400000 mismatches 0"), run.stdout[-1500:]


def test_lds_bank_model_of_the_shipped_lane_maps():
    """tools/lds_bank_model.py: the lane maps and pad periods compiled into decim_dense_kernel<8 / 16 / 32> and the
    interpolator's lane map + XOR swizzle are free of LDS bank conflicts under the service groups of
    MI355X_MICROARCH.md (a CPU model; the GPU counters that agree with it are in profiles/round3_*_summary.json)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "lds_bank_model.py")
    run = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.count("-> 0 extra LDS cycles") == 3
    # the maps in the tool are the maps in the kernel's header comment
    hdr = open(os.path.join(root, "sxxcvr_amd", "csrc", "sxfir_decim_dense.hip.h")).read()
    for D, bits in ((32, "c1, c2, g1, c0, p, g0"), (16, "g0, g2, g1, c1, p, c0"), (8, "g2, g3, g1, c0, p, g0")):
        assert re.search(r"D = +%d, pad per +\d+ rows: \(b0\.\.b5\) = \(%s\)" % (D, re.escape(bits)), hdr), D
        assert ("%d: (" % D) in open(tool).read()
    run = subprocess.run([sys.executable, tool, "interp"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0
    for L in (8, 16, 32):
        assert re.search(r"x%-2d extra LDS cycles per sub-tile: window reads 0, transposition writes 0, read-back 0" % L, run.stdout), run.stdout
