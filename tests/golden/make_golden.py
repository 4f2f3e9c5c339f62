#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Nothing here imports the oracle or the product: every expected value comes
from an INDEPENDENT route (numpy/scipy fp64, exact rational arithmetic, hand
evaluation of the reference's arithmetic, or the reference's own code), so the
fixtures can pin the oracle.  The reference itself (tejeez/sxxcvr) has no
tests, no vectors and no software FIR, and its translation unit as a whole
cannot be built in this image (SoapySDR + ALSA headers are absent).  The one
piece of per-sample arithmetic it owns CAN be compiled: convert_rx_buffer /
convert_tx_buffer (SoapySX.cpp:103-137) -- convert_kat.npz holds the output of
those two functions compiled from /root/reference (oracle/Makefile target
`ref`; needs this container).  Everything else comes from the other routes.

Run:  python tests/golden/make_golden.py      (rewrites the fixture files)
"""
import json
import os
from fractions import Fraction

import numpy as np
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# taps: Kaiser-windowed sinc, cutoff 0.5/ratio, sum = gain, fp64 -> fp32 once
# --------------------------------------------------------------------------
def design(ntaps, ratio, beta=8.0, gain=1.0):
    k = np.arange(ntaps, dtype=np.float64)
    mid = 0.5 * (ntaps - 1)
    fc = 0.5 / ratio
    h = 2.0 * fc * np.sinc(2.0 * fc * (k - mid)) * signal.windows.kaiser(ntaps, beta, sym=True)
    h *= gain / h.sum()
    return h.astype(np.float32)


def make_taps():
    out = {}
    for name, (n, r, g) in {
        "n128_d4": (128, 4, 1.0),
        "n256_d8": (256, 8, 1.0),
        "n256_l8": (256, 8, 8.0),
        "n1024_d32": (1024, 32, 1.0),
        # the reference's two slowest rates (SoapySX.cpp:180-208: dividers 768 and 1536 -> ratios 48 and 96), both directions
        "n1536_d48": (1536, 48, 1.0),
        "n3072_d96": (3072, 96, 1.0),
        "n1536_l48": (1536, 48, 48.0),
        "n3072_l96": (3072, 96, 96.0),
    }.items():
        out[name] = design(n, r, gain=g)
    np.savez(os.path.join(HERE, "taps.npz"), **out)
    return out


# --------------------------------------------------------------------------
# FIR known answers: scipy.signal.upfirdn in fp64 on fp32-representable data
# --------------------------------------------------------------------------
def make_fir(taps):
    rng = np.random.default_rng(0x51255)
    n = 1536
    x = (rng.integers(-(2 ** 23), 2 ** 23, size=n) / 2.0 ** 23
         + 1j * rng.integers(-(2 ** 23), 2 ** 23, size=n) / 2.0 ** 23).astype(np.complex64)
    out = {"x": x}
    for name, d in (("n128_d4", 4), ("n256_d8", 8), ("n1024_d32", 32)):
        h = taps[name].astype(np.float64)
        y = signal.upfirdn(h, x.astype(np.complex128), up=1, down=d)[: (n + d - 1) // d]
        out["decim_" + name] = y
    h = taps["n256_l8"].astype(np.float64)
    xs = x[:256]
    out["interp_n256_l8"] = signal.upfirdn(h, xs.astype(np.complex128), up=8, down=1)[: 256 * 8]
    # ratios 48 and 96 on a longer block (their filters are 1536 and 3072 taps long); a generator of its own, so that
    # the arrays above stay what they were
    rng2 = np.random.default_rng(0x51255 + 48)
    n2 = 12288
    x2 = (rng2.integers(-(2 ** 23), 2 ** 23, size=n2) / 2.0 ** 23
          + 1j * rng2.integers(-(2 ** 23), 2 ** 23, size=n2) / 2.0 ** 23).astype(np.complex64)
    out["x_long"] = x2
    for name, d in (("n1536_d48", 48), ("n3072_d96", 96)):
        h = taps[name].astype(np.float64)
        out["decim_" + name] = signal.upfirdn(h, x2.astype(np.complex128), up=1, down=d)[: (n2 + d - 1) // d]
    for name, l in (("n1536_l48", 48), ("n3072_l96", 96)):
        h = taps[name].astype(np.float64)
        out["interp_" + name] = signal.upfirdn(h, x2[:128].astype(np.complex128), up=l, down=1)[: 128 * l]
    # impulse + step edge cases (exact answers: the taps themselves / their prefix sums)
    np.savez(os.path.join(HERE, "fir_kat.npz"), **out)


# --------------------------------------------------------------------------
# time: exact round-half-away of ticks*1e9/rate and ns*rate/1e9
# --------------------------------------------------------------------------
def rnd(fr):
    s = 1 if fr >= 0 else -1
    fr = abs(fr)
    q = fr.numerator // fr.denominator
    if 2 * (fr - q) >= 1:
        q += 1
    return s * q


def make_time():
    clocks = [32.0e6, 38.4e6]
    divs = [1536, 768, 512, 256, 128, 64]            # SoapySX.cpp:196-208
    ticks = [0, 1, 255, 256, 768, 1792, 65536, 75000, 10 ** 9 + 7, 123456789012]
    rows = []
    for c in clocks:
        for d in divs:
            rate = c / d                                  # SoapySX.cpp:1205
            fr = Fraction(rate)                           # the double, exactly
            for t in ticks:
                ns = rnd(Fraction(t) * 10 ** 9 / fr)
                back = rnd(Fraction(ns) * fr / 10 ** 9)
                rows.append({"rate": rate, "ticks": t, "ns": ns, "ticks_back": back})
    hand = [  # round(256e9 / rate): SURVEY.md section 8 a-T
        {"rate": 600000.0, "ticks": 256, "ns": 426667},
        {"rate": 75000.0, "ticks": 256, "ns": 3413333},
        {"rate": 75000.0, "ticks": 512, "ns": 6826667},
        {"rate": 32.0e6 / 768, "ticks": 256, "ns": 6144000},
        # example/linear_repeater.py:40-43: 768 samples at 75 kS/s = 10.24 ms
        {"rate": 75000.0, "ticks": 768, "ns": 10240000},
    ]
    with open(os.path.join(HERE, "time_kat.json"), "w") as f:
        json.dump({"exact": rows, "hand": hand}, f, indent=0)


# --------------------------------------------------------------------------
# conversion: numpy float32 route + hand-evaluated edge values
# --------------------------------------------------------------------------
def conv_tx_numpy(x, thr2):
    f = x.view(np.float32).astype(np.float32)
    c = np.minimum(f, np.float32(1.0))                   # std::min(f, 1.0f)
    c = np.maximum(c, np.float32(-1.0))
    v = (c * np.float32(2147483648.0)).astype(np.float64)
    v = np.trunc(v)
    v = np.where(np.isnan(v), 0.0, v)
    v = np.clip(v, -2147483648.0, 2147483647.0).astype(np.int64)   # saturating (ARM) definition
    v = (v & ~3).astype(np.int64)
    fi, fq = f[0::2], f[1::2]
    mag = (fi * fi).astype(np.float32) + (fq * fq).astype(np.float32)
    ptt = mag.astype(np.float32) >= np.float32(thr2)
    v[0::2] |= np.where(ptt, 3, 0)
    return v.astype(np.int64)


REF_LIB = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libsxref.so")


def load_reference_converters():
    """oracle/_ref/libsxref.so: the reference's own convert_rx_buffer / convert_tx_buffer
    (SoapySX/SoapySX.cpp:103-137), compiled from /root/reference by `make -C oracle ref` (container only; those
    two functions need three standard headers and nothing else).  None when the reference is not here."""
    import ctypes as C
    import subprocess
    odir = os.path.dirname(os.path.dirname(REF_LIB))
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", odir, "-s", "ref"])
    if not os.path.exists(REF_LIB):
        return None
    lib = C.CDLL(REF_LIB)
    sz, vp = C.c_size_t, C.c_void_p
    lib.sxref_convert_rx_buffer.argtypes = [vp, sz, vp, sz, sz]
    lib.sxref_convert_tx_buffer.argtypes = [vp, sz, vp, sz, sz, C.c_float]
    lib.sxref_provenance.restype = C.c_char_p
    lib.sxref_init_register.restype = C.c_uint
    lib.sxref_sample_rate_row.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    return lib


def savez_deterministic(path, **arrays):
    """np.savez with fixed zip timestamps: the same arrays give the same bytes."""
    import zipfile
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        for name, a in arrays.items():
            info = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.external_attr = 0o644 << 16
            with z.open(info, "w", force_zip64=False) as f:
                np.lib.format.write_array(f, np.asanyarray(a), allow_pickle=False)


def make_conv():
    """convert_kat.npz = inputs + what the REFERENCE's own compiled converters return for them (x86-64 build of
    SoapySX.cpp:103-137), with one stated exception: rows of tx_in the C++ leaves undefined (a component that is
    >= 1.0 after the clamp, or NaN: the float -> int32 conversion at :124-125 overflows) carry
    the saturating definition (what the reference's AArch64 platform executes) from conv_tx_numpy instead, and
    tx_defined marks them 0.  The numpy route is asserted equal to the reference on every defined row, so the
    two routes pin each other."""
    import ctypes as C
    ref = load_reference_converters()
    if ref is None:
        print("convert_kat.npz / convert_hand.json kept as committed: /root/reference is not here, "
              "and their expected values are the reference's own output")
        return
    rng = np.random.default_rng(1255)
    n = 1 << 15                                                         # complex samples: 2^16 words each way
    s32 = rng.integers(-(2 ** 31), 2 ** 31, size=2 * n, dtype=np.int64).astype(np.int32)
    s32[:8] = [2 ** 31 - 1, 0x7FFFFF80, -(2 ** 31), 1, 0, -1, 0x7FFFFFBF, 0x7FFFFFC0]     # SURVEY.md 8 a-3 edges, RNE tie
    rx = np.empty(2 * n, dtype=np.float32)
    # offsets exercised too: the second half is converted by a call with src_offset = dest_offset = n / 2
    ref.sxref_convert_rx_buffer(s32.ctypes.data_as(C.c_void_p), 0, rx.ctypes.data_as(C.c_void_p), 0, n // 2)
    ref.sxref_convert_rx_buffer(s32.ctypes.data_as(C.c_void_p), n // 2, rx.ctypes.data_as(C.c_void_p), n // 2, n - n // 2)
    assert np.array_equal(rx, (s32.astype(np.float32) * np.float32(2.0 ** -31)).astype(np.float32)), \
        "numpy route and the reference's convert_rx_buffer disagree"
    tx_in = (rng.uniform(-1.2, 1.2, size=n) + 1j * rng.uniform(-1.2, 1.2, size=n)).astype(np.complex64)
    # a tenth of the block around the keying threshold (|v| ~ 1e-3), where fi*fi + fq*fq >= thr2 decides bit 0-1
    k = n // 10
    tx_in[n - k:] = (rng.uniform(-1.5e-3, 1.5e-3, size=k) + 1j * rng.uniform(-1.5e-3, 1.5e-3, size=k)).astype(np.complex64)
    below1 = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    edges = [0.5 + 0j, -0.25 + 0j, 1.0 + 1.0j, -1.0 - 1.0j, 2.0 - 3.0j, 1e-4 + 1e-4j, 0j, 0.001 + 0j,
             complex(below1, -below1), complex(-1.2, 0.999), complex(0.0, 1.0), complex(np.nan, 0.5),
             complex(6e-4, 8e-4), complex(float(np.float32(1e-3)), 0.0), complex(-0.0, 1e-3), complex(1e-30, -1e-30)]
    tx_in[:len(edges)] = np.array(edges, dtype=np.complex64)
    thr = np.float32(1.0e-3)
    thr2 = np.float32(thr * thr)                          # SoapySX.cpp:767-773
    tx_ref = np.empty(2 * n, dtype=np.int32)
    ref.sxref_convert_tx_buffer(tx_in.ctypes.data_as(C.c_void_p), 0, tx_ref.ctypes.data_as(C.c_void_p), 0, n // 2, thr2)
    ref.sxref_convert_tx_buffer(tx_in.ctypes.data_as(C.c_void_p), n // 2, tx_ref.ctypes.data_as(C.c_void_p), n // 2,
                                n - n // 2, thr2)
    f = tx_in.view(np.float32)
    comp_defined = ~(np.isnan(f) | (f >= np.float32(1.0)))            # per component: 2^31 * 1.0f does not fit int32
    defined = comp_defined[0::2] & comp_defined[1::2]
    sat = conv_tx_numpy(tx_in, thr2).astype(np.int32)
    rows = np.repeat(defined, 2)
    assert np.array_equal(sat[rows], tx_ref[rows]), "numpy route and the reference's convert_tx_buffer disagree"
    tx = np.where(rows, tx_ref, sat).astype(np.int32)
    prov = ref.sxref_provenance().decode() + "; g++ -O2 -ffp-contract=off, x86-64; via oracle/Makefile target ref"
    hand = {
        # exactly ldexpf((float)x, -31)
        "rx": [[2 ** 31 - 1, 1.0], [0x7FFFFF80, float.fromhex("0x1.fffffep-1")], [-(2 ** 31), -1.0],
               [1, 2.0 ** -31], [0, 0.0]],
        # SURVEY.md section 8 a-4 (threshold 1e-3): 0.5 -> 0x40000000|3, -0.25 -> 0xE0000000|3,
        # +1.0 saturates (build-defined, ARM behaviour) -> 0x7FFFFFFC, -1.0 -> 0x80000000
        "tx": [[0.5, 0.0, 0x40000003, 0], [-0.25, 0.0, 0xE0000003 - 2 ** 32, 0],
               [1.0, 1.0, 0x7FFFFFFF, 0x7FFFFFFC], [-1.0, -1.0, -(2 ** 31) + 3, -(2 ** 31)],
               [1e-4, 1e-4, 214748 & ~3, 214748 & ~3]],
    }
    # the hand values of the defined rows are what the reference returns, too
    for fi, fq, vi, vq in hand["tx"]:
        if fi < 1.0 and fq < 1.0:
            one = np.array([fi + 1j * fq], dtype=np.complex64)
            o = np.empty(2, dtype=np.int32)
            ref.sxref_convert_tx_buffer(one.ctypes.data_as(C.c_void_p), 0, o.ctypes.data_as(C.c_void_p), 0, 1, thr2)
            assert (int(o[0]), int(o[1])) == (vi, vq), (fi, fq, o)
    for sv, fv in hand["rx"]:
        one = np.array([sv, sv], dtype=np.int32)
        o = np.empty(2, dtype=np.float32)
        ref.sxref_convert_rx_buffer(one.ctypes.data_as(C.c_void_p), 0, o.ctypes.data_as(C.c_void_p), 0, 1)
        assert o[0] == np.float32(fv) and o[1] == np.float32(fv), (sv, o)
    savez_deterministic(os.path.join(HERE, "convert_kat.npz"), s32=s32, rx=rx, tx_in=tx_in, tx=tx,
                        tx_defined=defined.astype(np.uint8), thr2=np.array([thr2], dtype=np.float32),
                        provenance=np.frombuffer(prov.encode(), dtype=np.uint8))
    with open(os.path.join(HERE, "convert_hand.json"), "w") as f:
        json.dump(hand, f)
    print("convert_kat.npz: %d wire words / %d CF32 samples through the reference's converters (%s); "
          "%d tx rows undefined in the C++ carry the saturating definition" % (s32.size, n, prov, int((~defined).sum())))


# --------------------------------------------------------------------------
# sample-rate register table: the reference's own static data
# --------------------------------------------------------------------------
def make_rate_table():
    """rate_table.json = struct sampleRateRegs sample_rates[N_SAMPLE_RATES] (SoapySX.cpp:179-208) as the reference's
    compiled translation-unit fragment holds it: per supported rate the divider master clock / rate and the SX1255
    register fields setSampleRate programs (:1197-1203: 0x12 bits 3-0 = clkout, 0x13 bit 7 = mant, bit 6 = m, bits 5-3 = n).
    Checked here against the chip's own relation div = 8 * 3^m * 2^n."""
    import ctypes as C
    ref = load_reference_converters()
    if ref is None:
        print("rate_table.json kept as committed: /root/reference is not here")
        return
    rows = []
    for i in range(ref.sxref_n_sample_rates()):
        o = (C.c_uint * 5)()
        ref.sxref_sample_rate_row(i, o)
        div, clkout, mant, m, n = [int(v) for v in o]
        assert div == 8 * 3 ** m * 2 ** n and mant == 0, (div, m, n)
        rows.append({"div": div, "clkout": clkout, "mant": mant, "m": m, "n": n})
    # ... and the power-up register image init_registers[] (:139-176) the Device's register shadow starts from
    init = [int(ref.sxref_init_register(i)) for i in range(ref.sxref_n_init_registers())]
    assert len(init) == 0x14 and init[0x12] & 0x0F == 2 and (init[0x13] >> 3) & 7 == 5      # boots at div 256: the table's own row
    with open(os.path.join(HERE, "rate_table.json"), "w") as f:
        json.dump({"rows": rows, "init_registers": init,
                   "provenance": ref.sxref_provenance().decode() + "; via oracle/Makefile target ref"}, f, indent=0)
    print("rate_table.json: %d rows of sample_rates[] and %d init_registers[] from the reference (%s)"
          % (len(rows), len(init), ref.sxref_provenance().decode()))


# --------------------------------------------------------------------------
# stream rules: hand-evaluated traces of SoapySX.cpp:897-966 / :989-1104
# --------------------------------------------------------------------------
def make_stream():
    R = 75000.0
    rx = [
        # normal blocking reads of one period: timeNs = position/rate, flags = HAS_TIME (SX.cpp:950-953)
        {"pos": 0, "avail": 256, "period": 256, "buffer": 65536, "n": 256, "timeout": 100000, "rate": R,
         "exp": {"position": 256, "skipped": 0, "ret": 256, "flags": 4, "time_ns": 0}},
        {"pos": 256, "avail": 300, "period": 256, "buffer": 65536, "n": 256, "timeout": 100000, "rate": R,
         "exp": {"position": 512, "skipped": 0, "ret": 256, "flags": 4, "time_ns": 3413333}},
        {"pos": 512, "avail": 256, "period": 256, "buffer": 65536, "n": 256, "timeout": 100000, "rate": R,
         "exp": {"position": 768, "skipped": 0, "ret": 256, "flags": 4, "time_ns": 6826667}},
        # overrun: avail 70000 > 65536 -> overwritten 4464 -> (4464//256 + 2)*256 = 4864 skipped (SX.cpp:910-927)
        {"pos": 1000, "avail": 70000, "period": 256, "buffer": 65536, "n": 256, "timeout": 100000, "rate": R,
         "exp": {"position": 1000 + 4864 + 256, "skipped": 4864, "ret": 256, "flags": 4,
                 "time_ns": 78186667}},      # round((1000+4864)*1e9/75000) = 78186666.67
        # non-blocking: clamp to avail (SX.cpp:934-942)
        {"pos": 768, "avail": 100, "period": 256, "buffer": 65536, "n": 256, "timeout": 0, "rate": R,
         "exp": {"position": 868, "skipped": 0, "ret": 100, "flags": 4, "time_ns": 10240000}},
        {"pos": 768, "avail": 0, "period": 256, "buffer": 65536, "n": 256, "timeout": 0, "rate": R,
         "exp": {"position": 768, "skipped": 0, "ret": 0, "flags": 0, "time_ns": 0}},
        # period 1000 -> buffer 65000 (SX.cpp:464-466); avail 66001 -> overwritten 1001 -> 3 periods
        {"pos": 0, "avail": 66001, "period": 1000, "buffer": 65000, "n": 1000, "timeout": 1, "rate": R,
         "exp": {"position": 4000, "skipped": 3000, "ret": 1000, "flags": 4, "time_ns": 40000000}},
    ]
    tx = [
        # untimed, no underrun: continue at position (SX.cpp:1024-1038)
        {"pos": 1024, "avail": 64512, "delay": 1024, "period": 256, "n": 256, "flags": 0, "time_ns": 0,
         "timeout": 100000, "rate": R, "exp": {"position": 1280, "skipped": 0, "ret": 256, "discarded": 0}},
        # untimed underrun: delay -700 -> playback = pos+700 -> (700//256+2)*256 = 1024 forwarded
        {"pos": 2048, "avail": 66236, "delay": -700, "period": 256, "n": 256, "flags": 0, "time_ns": 0,
         "timeout": 100000, "rate": R, "exp": {"position": 2048 + 1024 + 256, "skipped": 1024, "ret": 256,
                                               "discarded": 0}},
        # timed: rx time 6826667 ns (pos 512) + 10.24 ms -> position 1280 exactly (linear_repeater.py:40-69)
        {"pos": 1024, "avail": 65536 - 200, "delay": 200, "period": 256, "n": 256, "flags": 4,
         "time_ns": 6826667 + 10240000, "timeout": 100000, "rate": R,
         "exp": {"position": 1536, "skipped": 256, "ret": 256, "discarded": 0}},
        # timed in the past: playback 2000-100 = 1900 > 1280 -> dropped, success reported (SX.cpp:1013-1023)
        {"pos": 2000, "avail": 65436, "delay": 100, "period": 256, "n": 256, "flags": 4,
         "time_ns": 6826667 + 10240000, "timeout": 100000, "rate": R,
         "exp": {"position": 2000, "skipped": 0, "ret": 256, "discarded": 1}},
        # timed, target behind position but not yet played: no rewind, written at position (posdiff <= 0)
        {"pos": 1500, "avail": 65036, "delay": 500, "period": 256, "n": 256, "flags": 4,
         "time_ns": 6826667 + 10240000, "timeout": 100000, "rate": R,
         "exp": {"position": 1756, "skipped": 0, "ret": 256, "discarded": 0}},
        # non-blocking clamp after a forward: local avail is reduced by the forward (SX.cpp:1072,1076-1085)
        {"pos": 0, "avail": 300, "delay": 65236, "period": 256, "n": 256, "flags": 4,
         "time_ns": 2666667, "timeout": 0, "rate": R,   # 200 samples
         "exp": {"position": 300, "skipped": 200, "ret": 100, "discarded": 0}},
    ]
    with open(os.path.join(HERE, "stream_kat.json"), "w") as f:
        json.dump({"rx": rx, "tx": tx}, f, indent=0)


if __name__ == "__main__":
    t = make_taps()
    make_fir(t)
    make_time()
    make_conv()
    make_rate_table()
    make_stream()
    print("golden fixtures written to", HERE)
