#!/usr/bin/env python3
"""Benchmark of the MI355X resampling hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3rx|3tx|5|5h|rx16|rx48|rx96|tx4|tx16|tx32|tx48|tx96]

One "step" = one streaming pass of the configuration's polyphase FIR (sxfir_decimate / sxfir_interpolate
through the C ABI: one kernel launch, history carry-over fused) over the rank's resident synthetic IQ block
(2^28 complex samples on the wideband side per GPU, already in HBM when the timed region starts);
consecutive steps are consecutive blocks of one continuous stream (filter history carried over on the GPU).

  --config 2   (default, the headline) 128-tap decimate-by-4, CF32          BASELINE config 2
  --config 3   full duplex: the 256-tap decimate-by-8 RX pass and the 256-tap interpolate-by-8 TX pass side by side
               on two HIP streams (BASELINE config 3), plus the timed readStream -> writeStream loop through the Device
  --config 3rx 256-tap decimate-by-8, CF32      --config 3tx  256-tap interpolate-by-8, CF32 (config 3's halves)
  --config 5   1024-tap decimate-by-32, CF32    --config 5h   the same with IQ stored as CF16 (config 5)

N = 1 runs one channel; N > 1 runs BASELINE config 4's layout (8 independent channels per GPU, 8*N in total,
no data-path collective) and measures the RCCL gather of the decimated output to rank 0 separately (reported
under "gather", never part of `value`: it is xGMI-link bound).  With N > 1 and no
torch.distributed environment this script starts the N ranks itself (fresh child processes, before the parent
touches a GPU); under torchrun it is one of the ranks.  Rank 0 prints ONE JSON line.

After the timed region the outputs of the last step are compared with the CPU oracle (start of the block, a
window in its middle, its end: the line carries "verified": true, and the script exits non-zero otherwise).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x51255
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                            # measured float4 copy ceiling (same guide)
VALU_PEAK_TFLOPS = 157.3                         # fp32 vector peak at 2.4 GHz (same guide)
XGMI_LINK_GBS = 153.0                            # one xGMI link, per direction (7 links per GPU, point to point)

# algorithmic bytes and flops per WIDEBAND-side sample (input of a decimator, output
# of the interpolator); halo and tap re-reads excluded.
CONFIGS = {
    "2": dict(mode="decim", ntaps=128, ratio=4, fmt="CF32", bytes=8 + 8 / 4, flop=128, gain=1.0,
              kernel="sxfir::decim4_wide_kernel<scalar taps, 8 outputs per lane>",
              name="128-tap polyphase decim-by-4, 1 ch CF32 streaming (BASELINE config 2)"),
    "3rx": dict(mode="decim", ntaps=256, ratio=8, fmt="CF32", bytes=8 + 8 / 8, flop=128, gain=1.0,
                kernel="sxfir::decim_dense_kernel<8, scalar taps: tap subsets on the four waves>",
                name="256-tap polyphase decim-by-8 RX, 1 ch CF32 streaming (BASELINE config 3, RX half)"),
    "3tx": dict(mode="interp", ntaps=256, ratio=8, fmt="CF32", bytes=8 + 8 / 8, flop=128, gain=8.0,
                kernel="sxfir::interp8_pass_kernel<2 inputs per lane, scalar taps>",
                name="256-tap polyphase interp-by-8 TX, 1 ch CF32 streaming (BASELINE config 3, TX half)"),
    "3": dict(mode="duplex", ntaps=256, ratio=8, fmt="CF32", bytes=9.0, flop=128, gain=1.0,
              kernel="sxfir::decim_dense_kernel<8> + sxfir::interp8_pass_kernel<2> on two streams",
              name="full-duplex 256-tap decim-8 RX + interp-8 TX (read+writeStream), timestamp-latency check "
                   "(BASELINE config 3)"),
    "5": dict(mode="decim", ntaps=1024, ratio=32, fmt="CF32", bytes=8 + 8 / 32, flop=128, gain=1.0,
              kernel="sxfir::decim_dense_kernel<32>",
              name="1024-tap decim-by-32, 1 ch CF32 streaming (BASELINE config 5, CF32 leg)"),
    "5h": dict(mode="decim", ntaps=1024, ratio=32, fmt="CF16", bytes=4 + 4 / 32, flop=128, gain=1.0,
               kernel="sxfir::decim_dense_kernel<32, CF16 storage: typed LDS-DMA converts half -> float on the way into the image>",
               name="1024-tap decim-by-32, 1 ch CF16 storage, fp32 arithmetic (BASELINE config 5, fp16 IQ leg)"),
}
# The reference's rate table (SoapySX.cpp:180-208: master clock / {64, 128, 256, 512, 768, 1536}; the converters run at master
# clock / 16, so ratio = divider / 16, 32 taps per phase): the rows the BASELINE configs above do not already cover.  Not
# bench lines of the driver: `--config rx48` etc. time, verify and profile these kernels the same way.
for _ratio, _rxk, _txk in ((4, None, "sxfir::interp8_pass_kernel<4 inputs per lane, x4: scalar taps, two passes>"), (16, "sxfir::decim_dense_kernel<16>", "sxfir::interp8_pass_kernel<2 inputs per lane, one phase block of sixteen>"),
                           (32, None, "sxfir::interp8_pass_kernel<2 inputs per lane, two phase blocks of sixteen>"),
                           (48, "sxfir::decim_blocks_kernel<3 blocks of 16 columns, scalar taps>", "sxfir::interp8_pass_kernel<2 inputs per lane, three phase blocks of sixteen>"),
                           (96, "sxfir::decim_blocks_kernel<6 blocks of 16 columns, scalar taps>", "sxfir::interp8_pass_kernel<2 inputs per lane, six phase blocks of sixteen>")):
    _khz = 38400.0 / 16 / _ratio
    if _rxk:
        CONFIGS["rx%d" % _ratio] = dict(mode="decim", ntaps=32 * _ratio, ratio=_ratio, fmt="CF32", bytes=8 + 8 / _ratio, flop=128, gain=1.0,
                                        kernel=_rxk, name="%d-tap decim-by-%d RX, 1 ch CF32 streaming (the reference's %g kS/s rate)"
                                        % (32 * _ratio, _ratio, _khz))
    CONFIGS["tx%d" % _ratio] = dict(mode="interp", ntaps=32 * _ratio, ratio=_ratio, fmt="CF32", bytes=8 + 8 / _ratio, flop=128,
                                    gain=float(_ratio), kernel=_txk,
                                    name="%d-tap interp-by-%d TX, 1 ch CF32 streaming (the reference's %g kS/s rate)"
                                    % (32 * _ratio, _ratio, _khz))


# ----------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves
# ----------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """Start n fresh child processes (one per GPU) running this script as torch.distributed ranks and wait for
    them.  Called before the parent has imported torch or made any GPU call; nothing is exec'ed over a process
    that has touched a GPU."""
    # the checker is built once, here, before the ranks exist (they would all find it stale at the same moment
    # after a fresh checkout; oracle_lib.build() also serialises itself with an flock).  CPU only: no GPU call.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SXFIR_BENCH_CHILD="1")
        # this pool's host driver supports dmabuf IPC only: with the legacy IPC mode RCCL's (and torch's) cross-process buffer
        # sharing fails with "hipIpcGetMemHandle: invalid argument".  The image exports the variable already; set here for a
        # caller whose environment lacks it, and recorded in the line (config.env) with the value the ranks really ran with
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, deadline = 0, None
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0:
                rc = rc or r
                if deadline is None:
                    deadline = time.time() + 30.0       # a rank died: give the others a moment, then stop them
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()                                 # exactly the processes started above
            for p in procs:
                p.wait()
            procs = []
            rc = rc or 1
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------------------------------------
# Board telemetry beside the kernel: socket power, its cap, the GFX clock the SMU reports
# ----------------------------------------------------------------------------------------------------------
class BoardSampler:
    """Side thread that reads the GPU's socket power and GFX clock every `period_s` while kernels run (amdsmi:
    the library behind `amd-smi metric --power --clock`; falls back to the hwmon files).  What it supplies is the
    driver-run evidence for DESIGN.md's "the package power cap sets the clock": power at the cap and a GFX clock
    far under the 2.4 GHz the chip runs unloaded.  Never raises: a box without telemetry yields {"error": ...}."""

    def __init__(self, gpu_index=0, period_s=0.02, bdf=None):
        import threading
        self.gpu_index, self.period, self.bdf = gpu_index, period_s, bdf
        self.samples, self.cap_w, self.source, self.error = [], None, None, None
        self._stop = threading.Event()
        self._ready = threading.Event()          # telemetry opened (or given up on): the sampled loop may start
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _open(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            h = handles[min(self.gpu_index, len(handles) - 1)]
            if self.bdf:
                # the HIP device index and the SMI's enumeration need not agree on a multi-GPU node: go by PCI address
                for cand in handles:
                    try:
                        if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower() == self.bdf.lower():
                            h = cand
                            break
                    except Exception:
                        pass
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h)["power_cap"]
                self.cap_w = cap / 1e6 if cap > 100000 else float(cap)
            except Exception:
                pass

            def read():
                pw = amdsmi.amdsmi_get_power_info(h)
                w = pw.get("current_socket_power")
                if not isinstance(w, (int, float)) or w <= 0:
                    w = pw.get("socket_power")
                if not isinstance(w, (int, float)) or w <= 0:
                    w = pw.get("average_socket_power")
                mhz = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)["clk"]
                return float(w), float(mhz)

            read()
            self.source = "amdsmi"
            return read
        except Exception as e:
            first = "%s: %s" % (type(e).__name__, e)
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
        if not cards:
            self.error = "no telemetry (amdsmi: %s; no hwmon power1_average)" % first
            return None
        d = os.path.dirname(cards[min(self.gpu_index, len(cards) - 1)])
        try:
            self.cap_w = int(open(os.path.join(d, "power1_cap")).read()) / 1e6
        except Exception:
            pass

        def read():
            w = int(open(os.path.join(d, "power1_average")).read()) / 1e6
            try:
                mhz = int(open(os.path.join(d, "freq1_input")).read()) / 1e6
            except Exception:
                mhz = float("nan")
            return w, mhz

        self.source = "hwmon"
        return read

    def _run(self):
        try:
            read = self._open()
        finally:
            self._ready.set()
        if read is None:
            return
        while not self._stop.is_set():
            try:
                self.samples.append(read() + (time.perf_counter(),))
            except Exception as e:
                self.error = "%s: %s" % (type(e).__name__, e)
                return
            self._stop.wait(self.period)

    def start(self, wait_s=20.0):
        # the first amdsmi_init() on a fresh box can take longer than the sampled loop runs (a driver-style run once
        # came back with "no samples"): wait until the telemetry is open, bounded
        self._thread.start()
        self._ready.wait(wait_s)

    def window(self, t0, t1):
        """Median socket power / GFX clock of the samples taken in [t0, t1] (time.perf_counter()), None without any."""
        sel = [s for s in list(self.samples) if t0 <= s[2] <= t1]
        if not sel:
            return None
        w = sorted(s[0] for s in sel)
        f = sorted(s[1] for s in sel)
        return {"power_w": round(w[len(w) // 2], 1), "gfx_mhz_smi": round(f[len(f) // 2], 0), "samples": len(sel)}

    def stop(self):
        self._stop.set()
        self._thread.join(5.0)
        if not self.samples:
            return {"error": self.error or "no samples"}
        w = sorted(s[0] for s in self.samples)
        f = sorted(s[1] for s in self.samples)
        return {"source": self.source, "samples": len(w), "power_w": round(w[len(w) // 2], 1),
                "power_w_min": round(w[0], 1), "power_w_max": round(w[-1], 1), "power_cap_w": self.cap_w,
                "gfx_mhz_smi": round(f[len(f) // 2], 0), "gfx_mhz_smi_min": round(f[0], 0), "gfx_mhz_smi_max": round(f[-1], 0)}


# ----------------------------------------------------------------------------------------------------------
# CPU side: oracle (checker and reported baseline)
# ----------------------------------------------------------------------------------------------------------
def load_oracle(fast=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()
    return oracle_lib.Oracle(fast=fast)


def cpu_baseline(cfg, seconds_target=10.0):
    """The build's own CPU FIR (the reference has none), order-matched fp32 oracle, timed on this host on a
    bounded sample of the same workload; plus the reference-equivalent conversion-only figure
    (convert_rx_buffer, SoapySX.cpp:103-112: all the per-sample arithmetic the reference's readStream does)."""
    import numpy as np
    cpuinfo = open("/proc/cpuinfo").read()
    fast = (" avx2" in cpuinfo) and (" fma" in cpuinfo)
    orc = load_oracle(fast=fast)
    ntaps, ratio = cfg["ntaps"], cfg["ratio"]
    taps = orc.design_lowpass(ntaps, ratio, 8.0, cfg["gain"])
    threads = orc.max_threads()
    model = "unknown"
    for line in cpuinfo.splitlines():
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    if cfg["mode"] == "decim":
        js, cw = 2, 4
        rot = 1 if ratio in (48, 96) else 0                     # the contract's rotation (sxfir_contract_rotation)

        def run(x, nthr):
            orc.decim_f32(taps, ratio, x, js, cw, threads=nthr, rot=rot)
        n_probe, n = 1 << 21, 1 << 26
        wide = lambda m: m                                      # wideband samples per call
    else:
        def run(x, nthr):
            orc.interp_f32_mt(taps, ratio, x, 2, threads=nthr)
        n_probe, n = (1 << 21) // ratio, (1 << 26) // ratio
        wide = lambda m: m * ratio
    x = orc.synth_iq(SEED, 0, 0, n_probe)
    t0 = time.perf_counter()
    run(x, 1)
    one_thread = wide(n_probe) / (time.perf_counter() - t0) / 1e6
    # bounded sample: the first 2^26 wideband samples of the workload's stream, filtered repeatedly
    x = orc.synth_iq_mt(SEED, 0, 0, n, threads)
    run(x, threads)                                             # warm-up (page faults, thread pool)
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds_target and reps < 1000:
        run(x, threads)
        reps += 1
        dt = time.perf_counter() - t0
    # conversion only: S32_LE wire words -> CF32, the reference's actual per-sample work
    m = 1 << 26
    words = np.frombuffer(np.random.default_rng(1).bytes(8 * m), dtype=np.int32)
    out = np.empty(2 * m, dtype=np.float32)
    orc.convert_rx_into(words, out, 1)
    t0 = time.perf_counter()
    orc.convert_rx_into(words, out, 1)
    conv1 = m / (time.perf_counter() - t0) / 1e6
    # ... and by several threads: the copy is memory bound and the buffers live on the NUMA node of the thread that
    # first touched them, so all hardware threads are not the fastest team; the sweep reports each and the best
    sweep = {}
    for nthr in sorted({4, 16, 64, threads} & set(range(1, threads + 1)) | {threads}):
        orc.convert_rx_into(words, out, nthr)
        creps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 0.6:
            orc.convert_rx_into(words, out, nthr)
            creps += 1
        sweep[nthr] = m * creps / (time.perf_counter() - t0) / 1e6
    convn = sweep[threads]
    best_threads = max(sweep, key=lambda k: sweep[k])
    if conv1 >= sweep[best_threads]:
        best_threads, best = 1, conv1
    else:
        best = sweep[best_threads]
    # ... and the REFERENCE'S OWN convert_rx_buffer, where its compiled form is here (oracle/_ref/libsxref.so: SoapySX.cpp:103-112
    # compiled in the build container between three standard headers; it travels to the GPU box as a prebuilt checker): one
    # thread, as the reference's readStream runs it (SoapySX.cpp:961)
    ref_conv = None
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libsxref.so")
    if os.path.exists(ref_lib):
        try:
            import ctypes as C
            ref = C.CDLL(ref_lib)
            ref.sxref_convert_rx_buffer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
            ref.sxref_provenance.restype = C.c_char_p
            ref.sxref_convert_rx_buffer(words.ctypes.data, 0, out.ctypes.data, 0, m)
            rreps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 0.6:
                ref.sxref_convert_rx_buffer(words.ctypes.data, 0, out.ctypes.data, 0, m)
                rreps += 1
            ref_conv = {"kind": "reference", "one_thread_MS/s": round(m * rreps / (time.perf_counter() - t0) / 1e6, 1),
                        "what": "the reference's own convert_rx_buffer, compiled from " + ref.sxref_provenance().decode()}
        except (OSError, AttributeError) as e:
            ref_conv = {"kind": "reference", "error": str(e)[:120]}
    return {
        "value": round(wide(n) * reps / dt / 1e6, 2),
        "unit": "MS/s (complex wideband-side samples)",
        "cores": threads,
        "kind": "port",
        "sample": "first %d wideband samples of the same synthetic channel-0 stream filtered %d times, %d threads "
                  "(OpenMP over output blocks), %s build; the reference has no software FIR, this is the "
                  "build's own CPU FIR" % (wide(n), reps, threads, "AVX2+FMA" if fast else "portable"),
        "one_thread_value": round(one_thread, 2),
        "cpu_model": model,
        "host_hw_threads": os.cpu_count(),                   # `cores` = what the affinity mask and the cgroup CPU quota leave of them
        "seconds": round(dt, 2),
        "conversion_only": {
            "what": "convert_rx_buffer (SoapySX.cpp:103-112), the reference's own per-sample readStream arithmetic: "
                    "S32_LE wire words -> CF32, oracle sxo_convert_rx",
            "one_thread_MS/s": round(conv1, 1),
            "all_threads_MS/s": round(convn, 1),
            "threads": threads,
            "by_threads_MS/s": {str(k): round(v, 1) for k, v in sorted(sweep.items())},
            "best_MS/s": round(best, 1),
            "best_threads": best_threads,
            "sample": "%d random wire-word samples converted repeatedly" % m,
            "reference_compiled": ref_conv,
        },
    }


def verify(cfg, plan, x, y, n_in, first_channel, history_from_block, orc, base=0):
    """Compare three windows of every channel of the output block y (its start, one around the middle -- tile
    seams of every kernel included --, its end) with the order-matched oracle; y's channels are the synthetic
    stream's channels first_channel, first_channel + 1, ... and the block's input is samples [base, base + n_in) of
    each.  history_from_block: the filter entered this block with the block's own tail as history (steady-state
    streaming over the same buffer), else with zero history (after a reset).  x is unused (the input is regenerated
    on the host from the seed).  Returns (ok, outputs compared)."""
    import numpy as np
    ratio, ntaps, decim = cfg["ratio"], cfg["ntaps"], cfg["mode"] == "decim"
    h = cfg["taps"] if cfg.get("taps") is not None else orc.design_lowpass(ntaps, ratio, 8.0, cfg["gain"])
    js, cw = plan.contract
    n_out = n_in // ratio if decim else n_in * ratio
    win = min(2048, n_out)
    starts = sorted(set([0, max(0, (n_out // 2 // 256) * 256 - win // 2), n_out - win]))
    compared, ok = 0, True
    yc = y if y.dim() == 2 else y.unsqueeze(0)
    for c in range(yc.shape[0]):
        ch = first_channel + c
        for o0 in starts:
            got = yc[c, o0:o0 + win].cpu().numpy()
            if decim:
                pad = ((ntaps + ratio - 1) // ratio) * ratio            # >= ntaps - 1, a multiple of the ratio
                s0 = o0 * ratio - pad                                   # first input sample of the window
                cnt = pad + win * ratio
            else:
                pad = ntaps // ratio                                    # input rows of history
                s0 = o0 // ratio - pad
                cnt = pad + (win + ratio - 1) // ratio + 1
            cnt = min(cnt, n_in - s0)
            if s0 < 0 and history_from_block:
                # steady-state streaming over the same buffer: the samples before the block are its own tail
                xw = np.concatenate([orc.synth_iq(SEED, ch, base + n_in + s0, -s0), orc.synth_iq(SEED, ch, base, cnt + s0)])
            elif s0 < 0:
                # after a reset: zeros before the block
                xw = np.concatenate([np.zeros(-s0, dtype=np.complex64), orc.synth_iq(SEED, ch, base, cnt + s0)])
            else:
                xw = orc.synth_iq(SEED, ch, base + s0, cnt)
            if cfg["fmt"] == "CF16":
                xw = orc.f16_to_f32(orc.f32_to_f16(xw.view(np.float32))).view(np.complex64)
            if decim:
                ref = orc.decim_f32(h, ratio, xw, js, cw, m0=pad // ratio, n_out=win, rot=plan.contract.rot)
            else:
                ref = orc.interp_f32(h, ratio, xw, js, n0=pad * ratio + (o0 % ratio), n_out=win)
            if cfg["fmt"] == "CF16":
                same = np.array_equal(got.view(np.uint16), orc.f32_to_f16(ref.view(np.float32)).ravel())
            else:
                same = np.array_equal(got.view(np.uint64), ref.view(np.uint64))
            ok = ok and bool(same)
            compared += win
    return ok, compared


def rate_table(orc, dev, log2n=28, launches=40):
    """Every rate of the reference's table (SoapySX.cpp:180-208: master clock / {64 .. 1536}; ratio = divider / 16, 32 taps per
    phase), RX and TX: each plan filters a block of the synthetic stream once from a reset and is checked against the oracle
    (three windows), then its kernel is timed over `launches` back-to-back passes with events on the launch stream.  Reported
    beside `value`, never part of it: fewer launches than the headline's and no settling phase, so a coarser figure."""
    import torch
    import sxxcvr_amd
    from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE
    rows, ok_all = {}, True
    for mode in ("decim", "interp"):
        for ratio in (4, 8, 16, 32, 48, 96):
            ntaps = 32 * ratio
            wide = (1 << log2n) - (1 << log2n) % (512 * ratio)
            cfg = dict(mode=mode, ntaps=ntaps, ratio=ratio, fmt="CF32", gain=1.0 if mode == "decim" else float(ratio))
            taps = sxxcvr_amd.design_lowpass(ntaps, ratio, 8.0, cfg["gain"])
            plan = sxxcvr_amd.Resampler(DECIMATE if mode == "decim" else INTERPOLATE, taps, ratio)
            n_in = wide if mode == "decim" else wide // ratio
            n_out = wide // ratio if mode == "decim" else wide
            x = torch.empty(n_in, dtype=torch.complex64, device=dev)
            sxxcvr_amd.synth_fill(x, SEED, first_channel=0, start=0, fmt="CF32")
            y = torch.empty(n_out, dtype=torch.complex64, device=dev)
            plan.process(x, out=y)
            torch.cuda.synchronize()
            ok, compared = verify(cfg, plan, x, y, n_in, 0, False, orc)
            ok_all = ok_all and ok
            for _ in range(30):
                plan.process(x, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                plan.process(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / launches
            byt = (8 + 8 / ratio) * wide
            rows[("rx" if mode == "decim" else "tx") + str(ratio)] = {
                "ms_per_2^28": round(ms * (1 << 28) / wide, 4), "frac_of_8TBs": round(byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "verified": bool(ok), "outputs_compared": compared}
            del plan, x, y
    return {"note": "every rate of the reference's table (ratio = divider / 16, 32 taps per phase), kernel time scaled to 2^28 wideband "
                    "samples from 2^%d-sample blocks, %d launches each; bit-exact windows against the oracle; never part of value" % (log2n, launches),
            "verified": ok_all, "rows": rows}


def size_curve(dev, log2_sizes=(18, 20, 22, 24)):
    """Kernel time per call of every rate of the reference's table at the call sizes the API issues (readStream / writeStream
    blocks are 256 .. 8192 samples, SoapySX.cpp:868-1105; the Device's chains batch them into passes of 2^14 .. 2^26.6 wideband
    samples): microseconds per call from the C loop of sxfir_time_* (HIP events on the launch stream, back-to-back launches) and
    what each call launches (sxfir_launch_geometry).  `x_of_ratio_32` = time per sample relative to /32 (RX) or x32 (TX) at the
    same size: the rates with the largest tiles (/48, /96, x48, x96) are the ones a small call leaves workgroup slots empty for.
    Reported beside `value`, never part of it (tools/sizebench.py is the long form)."""
    import torch
    import sxxcvr_amd
    from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE
    rows = {}
    st = torch.cuda.current_stream(dev).cuda_stream
    for mode in ("decim", "interp"):
        for ratio in (4, 8, 16, 32, 48, 96):
            taps = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, 1.0 if mode == "decim" else float(ratio))
            plan = sxxcvr_amd.Resampler(DECIMATE if mode == "decim" else INTERPOLATE, taps, ratio)
            key = ("rx" if mode == "decim" else "tx") + str(ratio)
            rows[key] = {}
            for lg in log2_sizes:
                wide = (1 << lg) // ratio * ratio
                n_in = wide if mode == "decim" else wide // ratio
                n_out = wide // ratio if mode == "decim" else wide
                x = torch.empty(n_in, dtype=torch.complex64, device=dev)
                sxxcvr_amd.synth_fill(x, SEED, first_channel=0, start=0, fmt="CF32")
                y = torch.empty(n_out, dtype=torch.complex64, device=dev)
                g = plan.geometry(n_in)
                ms = plan.time_passes_ptr(x.data_ptr(), n_in, n_in, y.data_ptr(), n_out, 5, st)
                iters = max(10, min(1000, int(40.0 / max(ms, 1e-3))))
                ms = min(plan.time_passes_ptr(x.data_ptr(), n_in, n_in, y.data_ptr(), n_out, iters, st) for _ in range(2))
                rows[key]["2^%d" % lg] = {"us_per_call": round(ms * 1e3, 2), "tiles": g["n_tiles"], "items_per_tile": g["split"],
                                          "workgroups": g["workgroups"], "slots": g["resident"], "kernel": g["kernel"]}
                del x, y
            plan.close()
    for key, by_size in rows.items():
        base = rows[("rx" if key.startswith("rx") else "tx") + "32"]
        for sz, r in by_size.items():
            r["x_of_ratio_32"] = round(r["us_per_call"] / base[sz]["us_per_call"], 2)
    return {"note": "kernel time per call at the call sizes the API issues (wideband samples per call); sxfir_time_* C loop, HIP events on "
                    "the launch stream; below ~2^20 samples a call is one launch (the runtime's launch rate); never part of value",
            "rows": rows}


def through_device():
    """API-parity figures through the SoapySDR-style Device (readStream / writeStream incl. PCIe, staging and
    launch overheads), decimate-by-4 / interpolate-by-4 at 600 kS/s; never part of `value`."""
    import numpy as np
    import sxxcvr_amd.soapy as SoapySDR
    res = {"note": "through SoapySX-style readStream/writeStream (host buffers, PCIe, pageable caller memory); "
                   "API-parity figures (median call; 256-sample calls: mean), never part of value"}
    import sxxcvr_amd
    for blk, key, pin in ((256, "256_sample_calls", False), (65536, "65536_sample_calls", False),
                          (1 << 20, "1048576_sample_calls", False), (1 << 20, "1048576_sample_calls_registered_buffer", True)):
        dev = SoapySDR.Device({"driver": "sx", "clock": "virtual"})
        dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 600000.0)
        rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CF32", [0], {"period": str(min(blk, 65536))})
        tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, "CF32", [0], {"period": str(min(blk, 65536))})
        dev.activateStream(rx)
        dev.activateStream(tx)
        buf = np.zeros(blk, dtype=np.complex64)
        if pin:
            sxxcvr_amd.pin_array(buf)                    # page-locked: the decimator stores straight into it
        try:
            n = max(24, min(2000, (1 << 24) // blk))

            def per_call(fn):
                # median of the per-call times: a short loop's mean is at the mercy of one slow call
                fn()
                fn()
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    r = fn()
                    ts.append(time.perf_counter() - t0)
                    if r.ret != blk:
                        raise RuntimeError("stream call returned %d" % r.ret)
                ts.sort()
                return ts[len(ts) // 2] if blk >= 4096 else sum(ts) / len(ts)

            dt_rx = per_call(lambda: dev.readStream(rx, [buf], blk))
            dt_tx = per_call(lambda: dev.writeStream(tx, [buf], blk))
        finally:
            if pin:
                sxxcvr_amd.unpin_array(buf)
        res[key] = {"readStream_us_per_call": round(dt_rx * 1e6, 2), "readStream_out_MS/s": round(blk / dt_rx / 1e6, 1),
                    "readStream_wideband_in_MS/s": round(4 * blk / dt_rx / 1e6, 1),
                    "writeStream_us_per_call": round(dt_tx * 1e6, 2), "writeStream_in_MS/s": round(blk / dt_tx / 1e6, 1)}
        dev.deactivateStream(rx)
        dev.deactivateStream(tx)
        dev.closeStream(rx)
        dev.closeStream(tx)
    res["c_caller"] = c_caller()
    return res


def c_caller():
    """The same calls from C (sxxcvr_amd/lib/sx_devloop = tools/devloop.c against include/sx_device.h, built by
    build()): what the plugin costs per readStream / writeStream call with no Python stand-in in the loop, at the
    reference's block sizes (256 ... 8192, SoapySX.cpp:706-707) on the virtual sample clock.  A child process."""
    exe = os.path.join(ROOT, "sxxcvr_amd", "lib", "sx_devloop")
    if not os.path.exists(exe):
        return {"error": "sx_devloop is not built (python -m sxxcvr_amd.build)"}
    try:
        run = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=300)
        if run.returncode != 0:
            return {"error": (run.stdout + run.stderr)[-300:]}
        out = json.loads(run.stdout.strip().splitlines()[-1])
        out["note"] = "C caller through the flat Device API, virtual sample clock; mean over `calls` calls per block size"
        return out
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def gpu_identity(gpu_index):
    """What tells this rank's GPU from the others' on the node: PCI address, name, arch (C ABI, include/sxfir.h)."""
    import ctypes as C
    import sxxcvr_amd
    lib = sxxcvr_amd.load_sxfir()
    out = {"gpu_index": gpu_index}
    bdf = C.create_string_buffer(32)
    if lib.sxfir_device_pci_bus_id(gpu_index, bdf, 32) == 0:
        out["pci_bus_id"] = bdf.value.decode()
    name, arch, cus, hbm = C.create_string_buffer(64), C.create_string_buffer(32), C.c_int(), C.c_size_t()
    if lib.sxfir_device_info(gpu_index, name, arch, C.byref(cus), C.byref(hbm)) == 0:
        out.update(name=name.value.decode(), arch=arch.value.decode(), compute_units=cus.value)
    return out


class GatherCertifier:
    """Makes a gather's result checkable by the line that times it.  Every rank states two 64-bit checksums per
    channel of the block it is about to send (sxxcvr_amd.dist.block_checksums, computed on its GPU from the bits the
    kernel wrote); the tables are all-gathered over the host-side control group; rank 0 recomputes them over the
    gathered tensor -- every peer's block, channel by channel, in global channel order -- and compares the first channel
    of every rank's block with the CPU oracle on top (the oracle ties the bits to the filter, the checksums tie every
    other channel and sample to what its sender held).  A verified gather always carries a block of the stream that no
    earlier gather carried (next_block), so a stale destination cannot pass.

    SXFIR_BENCH_CORRUPT_RANK=r (test hook): rank r flips one bit of what it sends after stating its checksums."""

    def __init__(self, cfg, plan, orc, world, rank, ctl, total_channels, nchan_local, n_in):
        self.cfg, self.plan, self.orc = cfg, plan, orc
        self.world, self.rank, self.ctl = world, rank, ctl
        self.total, self.local, self.n_in = total_channels, nchan_local, n_in
        c = os.environ.get("SXFIR_BENCH_CORRUPT_RANK")
        self.corrupt = c is not None and c != "" and int(c) == rank

    def state(self, block):
        """Sender side, on the stream that produced `block`: the checksums of what is about to be sent (a device
        tensor; exchanged later, off the timed path).  With the corrupt hook: then damage the block."""
        import torch
        import sxxcvr_amd.dist as sxdist
        sums = sxdist.block_checksums(block)
        if self.corrupt:
            w = torch.view_as_real(block).view(torch.int64) if block.is_complex() else block.view(torch.int64)
            w[w.shape[0] // 2, w.shape[1] // 3] ^= 1 << 20
        return sums

    def check(self, full, sums, base, history_from_block):
        """Collective over the control group.  full: the gathered [total_channels, n_out] tensor on rank 0 (None
        elsewhere); sums: this rank's state() of the block it sent.  Returns the record for the JSON line (rank 0)."""
        import sxxcvr_amd.dist as sxdist
        stated = sxdist.exchange_checksums(sums, self.total, group=self.ctl)
        if self.rank != 0:
            return None
        bad = sxdist.check_gathered(full, stated)
        ok_o, n_o = True, 0
        for r in range(self.world):                        # the first channel of every rank's block against the oracle
            ch = r * self.local
            ok, n = verify(self.cfg, self.plan, None, full[ch:ch + 1], self.n_in, ch, history_from_block, self.orc, base=base)
            ok_o, n_o = ok_o and ok, n_o + n
        return {"verified": (not bad) and ok_o, "peer_blocks_checked": self.world - 1,
                "channels_checksummed": self.total, "bad_channels": bad,
                "oracle_outputs_on_first_channel_of_every_block": n_o, "oracle_ok": ok_o,
                "stream_block": base // self.n_in,
                "how": "per channel: sum and position-weighted sum of the 64-bit sample words mod 2^64, stated by the sender "
                       "from its own output, all-gathered over the host-side group, recomputed by rank 0 over the gathered "
                       "tensor; plus oracle windows on the first channel of every rank's block"}


def measure_gather(world, y, total_channels, host_collectives, cdev, wide_per_gpu, elapsed, steps, cert, next_block,
                   step_into):
    """Exchange step of BASELINE config 4: RCCL gather of every rank's decimated output to rank 0 over xGMI,
    timed after (and outside) the timed region; then one more gather of a block of the stream no earlier gather
    carried, certified by GatherCertifier (checksums of every peer block, oracle windows)."""
    import torch
    import torch.distributed as dist
    import sxxcvr_amd.dist as sxdist
    yg = y.cpu() if host_collectives else y
    full = None
    for _ in range(2):
        full = sxdist.gather_channels(yg, total_channels, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    g0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        full = sxdist.gather_channels(yg, total_channels, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    g = (time.perf_counter() - g0) / reps
    t = torch.tensor([g], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    g = float(t.item())
    peer_bytes = y.numel() * 8
    out = {
        "ms": round(g * 1e3, 3),
        "bytes_per_peer": peer_bytes,
        "GB/s_into_root": round(peer_bytes * (world - 1) / g / 1e9, 2),
        "GB/s_per_link": round(peer_bytes / g / 1e9, 2),
        "value_with_gather": round(world * wide_per_gpu * 1.0 / (elapsed / steps + g) / 1e6, 1),
        "note": "gather of the decimated output is xGMI per-link bound (~153 GB/s per peer) and not part of value",
    }
    # the certified gather: the stream's next block (zero history), stated by every sender, recomputed by the root
    del full
    base = next_block()
    step_into(y)
    sums = cert.state(y)
    yg = y.cpu() if host_collectives else y
    full = sxdist.gather_channels(yg, total_channels, dst=0)
    torch.cuda.synchronize()
    rec = cert.check(full, sums, base, False)
    out["rccl_ranks"] = dist.get_world_size()                    # as the communicator reports it
    if dist.get_rank() == 0:
        local = y.shape[0]
        out["root_holds_own_channels"] = bool(torch.equal(full[:local].to(y.device), y))
        out["gathered_shape"] = list(full.shape)
        out["check"] = rec
    return out


def measure_gather_capi(world, rank, gpu_index, y, total_channels, cdev, wide_per_gpu, elapsed, steps, step_into, cert,
                        next_block, psteps=6):
    """--gather capi: the same exchange through the C ABI (sxfir_comm_* over librccl, include/sxfir.h), the way a
    C / C++ host would run it: no torch.distributed in the data path.  torch.distributed only carries the 128-byte
    communicator id from rank 0 to the others (any launcher's store would do) and the barriers of the timing.
    Serial form, then the steady state: two output buffers in turn, each step's gather queued in 4 pieces (whole
    channels) on a side stream behind an event of the kernel that produced the block."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    import sxxcvr_amd
    lib = sxxcvr_amd.load_sxfir()

    def ck(rc, what):
        if rc != 0:
            raise RuntimeError("%s: %d %s" % (what, rc, lib.sxfir_last_error().decode("utf-8", "replace")))

    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        ck(lib.sxfir_comm_unique_id(buf), "sxfir_comm_unique_id")
        ident = torch.tensor(list(buf), dtype=torch.uint8)
    ident = ident.to(cdev)
    dist.broadcast(ident, src=0)
    raw = (C.c_ubyte * 128)(*ident.cpu().tolist())
    comm = C.c_void_p()
    ck(lib.sxfir_comm_init_rank(C.byref(comm), raw, world, rank, gpu_index), "sxfir_comm_init_rank")
    q_rank, q_n, q_dev = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    ck(lib.sxfir_comm_query(comm, C.byref(q_rank), C.byref(q_n), C.byref(q_dev)), "sxfir_comm_query")
    if (q_rank.value, q_n.value, q_dev.value) != (rank, world, gpu_index):
        raise RuntimeError("RCCL reports rank %d of %d on GPU %d; this process is rank %d of %d on GPU %d"
                           % (q_rank.value, q_n.value, q_dev.value, rank, world, gpu_index))
    peer_bytes = y.numel() * 8
    full = torch.empty((total_channels,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device) if rank == 0 else None
    recv = full.data_ptr() if rank == 0 else None
    side = torch.cuda.Stream(y.device)
    chunk = peer_bytes // 4

    def gather(src, stream):
        ck(lib.sxfir_comm_gather(comm, src.data_ptr(), recv, peer_bytes, peer_bytes, 0, chunk, stream), "sxfir_comm_gather")

    main_stream = torch.cuda.current_stream(y.device)
    for _ in range(2):
        gather(y, main_stream.cuda_stream)
    torch.cuda.synchronize()
    dist.barrier()
    g0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        gather(y, main_stream.cuda_stream)
    torch.cuda.synchronize()
    dist.barrier()
    g = (time.perf_counter() - g0) / reps
    t = torch.tensor([g], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    g = float(t.item())
    out = {
        "via": "C ABI: sxfir_comm_gather over librccl (ncclGroupStart; root ncclRecv x %d; peers ncclSend; ncclGroupEnd; "
               "4 pieces of whole channels)" % (world - 1),
        "ms": round(g * 1e3, 3), "bytes_per_peer": peer_bytes,
        "GB/s_into_root": round(peer_bytes * (world - 1) / g / 1e9, 2), "GB/s_per_link": round(peer_bytes / g / 1e9, 2),
        "value_with_gather": round(world * wide_per_gpu * 1.0 / (elapsed / steps + g) / 1e6, 1),
        "note": "gather of the decimated output is xGMI per-link bound (~153 GB/s per peer) and not part of value",
    }
    out["rccl_ranks"] = q_n.value                                # ncclCommCount of the communicator the gather ran on
    # the certified gather (serial form): the stream's next block, stated by every sender, recomputed by the root
    base = next_block()
    step_into(y)
    sums = cert.state(y)
    if rank == 0:
        full.zero_()
    gather(y, main_stream.cuda_stream)
    torch.cuda.synchronize()
    rec = cert.check(full, sums, base, False)
    if rank == 0:
        out["root_holds_own_channels"] = bool(torch.equal(full[: y.shape[0]], y))
        out["gathered_shape"] = list(full.shape)
        out["check"] = rec
    # steady state: kernel of step s on the main stream into buffer s % 2; its gather on the side stream behind an
    # event; before a buffer is overwritten the main stream waits for the gather that last read it
    ybuf = [y, torch.empty_like(y)]
    produced = [torch.cuda.Event(), torch.cuda.Event()]
    gathered = [None, None]
    stated = {}

    def run(nsteps, state_last=False):
        for s_ in range(nsteps):
            k = s_ % 2
            if gathered[k] is not None:
                main_stream.wait_event(gathered[k])
            step_into(ybuf[k])
            if state_last and s_ == nsteps - 1:
                stated["sums"] = cert.state(ybuf[k])
            produced[k].record(main_stream)
            side.wait_event(produced[k])
            gather(ybuf[k], side.cuda_stream)
            gathered[k] = torch.cuda.Event()
            gathered[k].record(side)
        torch.cuda.synchronize()

    run(2)
    dist.barrier()
    p0 = time.perf_counter()
    run(psteps)
    dist.barrier()
    tp = time.perf_counter() - p0
    t = torch.tensor([tp], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tp = float(t.item())
    per_link = peer_bytes * psteps / tp / 1e9
    out["overlapped_value"] = round(world * wide_per_gpu * psteps / tp / 1e6, 1)
    out["link_bound_frac"] = round(per_link / XGMI_LINK_GBS, 4)
    out["overlapped"] = {"steps": psteps, "ms_per_step": round(tp / psteps * 1e3, 3), "chunks_per_step": 4,
                         "GB/s_per_link": round(per_link, 2), "link_peak_GB/s": XGMI_LINK_GBS}
    # the certified steady state: two more steps on the stream's next block; the second one's gather is checked
    base = next_block()
    if rank == 0:
        full.zero_()
        torch.cuda.synchronize()
    run(2, state_last=True)
    rec = cert.check(full, stated["sums"], base, True)
    if rank == 0:
        out["overlapped"]["root_holds_own_channels"] = bool(torch.equal(full[: y.shape[0]], ybuf[1]))
        out["overlapped"]["check"] = rec
    lib.sxfir_comm_destroy(comm)
    return out


def measure_gather_pipelined(world, y, total_channels, cdev, wide_per_gpu, step_into, cert, next_block, psteps=6):
    """The same exchange in steady state; returns the keys to add to the line's "gather" object."""
    import torch
    import torch.distributed as dist
    import sxxcvr_amd.dist as sxdist
    out = {}
    peer_bytes = y.numel() * 8

    # The same exchange in steady state: each step's output is gathered in chunks of whole channels
    # behind the kernel that produced it and beside the kernels of the following steps (two output buffers in turn).
    # The throughput of that pipeline is max(kernel, gather) per step instead of their sum.
    depth = 2
    ybuf = [y, torch.empty_like(y)]
    pipe = sxdist.GatherPipeline(total_channels, tuple(y.shape), y.dtype, y.device, dst=0, chunks=4, depth=depth)

    stated = {}

    def run(nsteps, state_last=False):
        for s in range(nsteps):
            k = s % depth
            pipe.reuse(k)
            step_into(ybuf[k])
            if state_last and s == nsteps - 1:
                stated["sums"] = cert.state(ybuf[k])
            pipe.submit(k, ybuf[k])
        pipe.drain()

    run(depth)                                                   # warm-up: communicators, staging buffers
    torch.cuda.synchronize()
    dist.barrier()
    p0 = time.perf_counter()
    run(psteps)
    torch.cuda.synchronize()
    dist.barrier()
    tp = time.perf_counter() - p0
    t = torch.tensor([tp], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tp = float(t.item())
    per_link = peer_bytes * psteps / tp / 1e9
    out["overlapped_value"] = round(world * wide_per_gpu * psteps / tp / 1e6, 1)
    out["link_bound_frac"] = round(per_link / XGMI_LINK_GBS, 4)
    out["overlapped"] = {
        "steps": psteps, "ms_per_step": round(tp / psteps * 1e3, 3), "chunks_per_step": pipe.chunks,
        "GB/s_per_link": round(per_link, 2), "link_peak_GB/s": XGMI_LINK_GBS,
        "how": "each step's output gathered in %d chunks (whole channels) behind its kernel and beside the next "
               "steps' kernels, two output buffers in turn; value = whole-job MS/s of that pipeline" % pipe.chunks,
    }
    # the certified steady state: `depth` more steps on the stream's next block; the last one's gather is checked
    base = next_block()
    if dist.get_rank() == 0:
        for k in range(depth):
            pipe.full[k].zero_()
    torch.cuda.synchronize()
    run(depth, state_last=True)
    last = (depth - 1) % depth
    rec = cert.check(pipe.slot(last), stated["sums"], base, depth > 1)
    if dist.get_rank() == 0:
        out["overlapped"]["root_holds_own_channels"] = bool(
            torch.equal(pipe.slot(last)[: y.shape[0]].to(y.device), ybuf[last]))
        out["overlapped"]["gathered_shape"] = list(pipe.slot(last).shape)
        out["overlapped"]["check"] = rec
    return out


# ----------------------------------------------------------------------------------------------------------
# BASELINE config 3: full duplex
# ----------------------------------------------------------------------------------------------------------
def duplex_timed_loop(blocks=(256, 1024, 4096), latency_periods=3, rounds=400):
    """The reference's full-duplex pattern (example/linear_repeater.py:40-69) through the Device at decimate-by-8 /
    interpolate-by-8 (300 kS/s with decim=auto / interp=auto): every RX block is retransmitted with
    HAS_TIME at timeNs + latency; after every writeStream the TX position must be exactly
    rx_position + latency (SoapySX.cpp:950, :1012).  Reports microseconds per round trip; never part of value."""
    import numpy as np
    import sxxcvr_amd.soapy as SoapySDR
    rate = 300000.0
    out = {"note": "readStream -> writeStream(HAS_TIME, timeNs + latency) through the Device at /8 and x8, virtual sample "
                   "clock; TX_POSITION == rx_position + latency asserted on every block", "sample_rate": rate}
    for blk in blocks:
        dev = SoapySDR.Device({"driver": "sx", "clock": "virtual", "decim": "auto", "interp": "auto"})
        dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, rate)
        dev.setSampleRate(SoapySDR.SOAPY_SDR_TX, 0, rate)
        assert int(dev.readSetting("RX_DECIM")) == 8 and int(dev.readSetting("TX_INTERP")) == 8
        rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CF32", [0], {"period": str(blk)})
        tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, "CF32", [0], {"period": str(blk), "threshold": "0"})
        dev.activateStream(rx)
        dev.activateStream(tx)
        latency = latency_periods * blk
        dt_ns = SoapySDR.ticksToTimeNs(latency, rate)
        buf = np.zeros(blk, dtype=np.complex64)
        n_bad, ts = 0, []
        for i in range(rounds + 20):
            t0 = time.perf_counter()
            r = dev.readStream(rx, [buf], blk)
            w = dev.writeStream(tx, [buf], blk, flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=r.timeNs + dt_ns)
            t1 = time.perf_counter()
            if r.ret != blk or w.ret != blk:
                raise RuntimeError("stream call returned %d / %d" % (r.ret, w.ret))
            rx_pos = int(dev.readSetting("RX_POSITION"))
            if int(dev.readSetting("TX_POSITION")) != rx_pos + latency:
                n_bad += 1
            if SoapySDR.timeNsToTicks(r.timeNs, rate) != rx_pos - blk:
                n_bad += 1
            if i >= 20:
                ts.append(t1 - t0)
        ts.sort()
        out["%d_sample_blocks" % blk] = {"round_trip_us_median": round(ts[len(ts) // 2] * 1e6, 2),
                                         "round_trip_us_p99": round(ts[int(len(ts) * 0.99)] * 1e6, 2),
                                         "latency_samples": latency, "latency_ns": dt_ns, "rounds": rounds,
                                         "blocks_off_position": n_bad}
        out["latency_check_passed"] = out.get("latency_check_passed", True) and n_bad == 0
        for st in (rx, tx):
            dev.deactivateStream(st)
            dev.closeStream(st)
    return out


def main_duplex(args):
    """--config 3 (N = 1): one step = one decimate-by-8 pass over 2^28 wideband samples on the RX stream AND one
    interpolate-by-8 pass producing 2^28 wideband samples on the TX stream, launched side by side; value = wideband
    samples of both directions per second.  Per direction: HIP events on its own stream around the K timed steps."""
    import torch
    import sxxcvr_amd
    from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, StreamTimer
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.gpus != 1:
        raise SystemExit("--config 3 is a single-GPU workload (BASELINE config 3)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 1 << args.log2_samples
    rxc, txc = CONFIGS["3rx"], CONFIGS["3tx"]
    rx_plan = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(256, 8, 8.0, rxc["gain"]), 8)
    tx_plan = sxxcvr_amd.Resampler(INTERPOLATE, sxxcvr_amd.design_lowpass(256, 8, 8.0, txc["gain"]), 8)
    xr = torch.empty((1, n), dtype=torch.complex64, device=dev)
    yr = torch.empty((1, n // 8), dtype=torch.complex64, device=dev)
    xt = torch.empty((1, n // 8), dtype=torch.complex64, device=dev)
    yt = torch.empty((1, n), dtype=torch.complex64, device=dev)
    sxxcvr_amd.synth_fill(xr, SEED, first_channel=0, start=0)
    sxxcvr_amd.synth_fill(xt, SEED, first_channel=0, start=0)
    torch.cuda.synchronize()
    s_rx, s_tx = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def step():
        rx_plan.process_ptr(xr.data_ptr(), n, n, yr.data_ptr(), n // 8, s_rx.cuda_stream)
        tx_plan.process_ptr(xt.data_ptr(), n // 8, n // 8, yt.data_ptr(), n, s_tx.cuda_stream)

    for _ in range(args.settle + args.warmup):
        step()
    torch.cuda.synchronize()
    t_rx, t_tx = StreamTimer(s_rx.cuda_stream), StreamTimer(s_tx.cuda_stream)
    t0 = time.perf_counter()
    t_rx.start()
    t_tx.start()
    for _ in range(args.steps):
        step()
    t_rx.stop()
    t_tx.stop()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rx_ms, tx_ms = t_rx.elapsed_ms() / args.steps, t_tx.elapsed_ms() / args.steps

    orc = load_oracle()
    ok_rx, n_rx = verify(rxc, rx_plan, xr, yr, n, 0, True, orc)
    ok_tx, n_tx = verify(txc, tx_plan, xt, yt, n // 8, 0, True, orc)
    # each direction alone, same buffers, for the price of sharing the chip
    alone = {}
    for name, plan, a, b, ni, no in (("rx", rx_plan, xr, yr, n, n // 8), ("tx", tx_plan, xt, yt, n // 8, n)):
        plan.reset()
        torch.cuda.synchronize()
        alone[name] = plan.time_passes_ptr(a.data_ptr(), ni, ni, b.data_ptr(), no, min(max(args.steps, 40), 200),
                                           torch.cuda.current_stream(dev).cuda_stream)
    sampler = BoardSampler(0, 0.02)
    sampler.start()
    t_tel = time.perf_counter()
    while time.perf_counter() - t_tel < 1.0:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    board = sampler.stop()
    verified = ok_rx and ok_tx
    ms_per_step = elapsed / args.steps * 1e3
    value = 2 * n * args.steps / elapsed / 1e6
    bytes_step = 9.0 * n * 2

    def direction(ms, kernel):
        ach = 9.0 * n / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "span_ms_per_step": round(ms, 4), "achieved": round(ach, 1),
                "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(9.0 * n)}

    achieved = bytes_step / (ms_per_step * 1e-3) / 1e9
    line = {
        "metric": "complex MS/s, full-duplex 256-tap decim-by-8 RX + interp-by-8 TX CF32 (wideband rate, both directions)",
        "value": round(value, 1), "unit": "MS/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "verified": verified,
        "config": {"workload": "1xMI355X: " + CONFIGS["3"]["name"], "bench_config": "3", "ntaps": 256, "decim": 8, "interp": 8,
                   "format": "CF32", "wideband_samples_per_direction": n, "untimed_settle_launches": args.settle,
                   "verified_outputs": n_rx + n_tx,
                   "step": "one /8 pass over 2^%d wideband samples on the RX stream and one x8 pass producing 2^%d on the "
                           "TX stream, side by side" % (args.log2_samples, args.log2_samples)},
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": CONFIGS["3"]["kernel"],
            "algorithmic_bytes_per_step": int(bytes_step),
            "how": "both directions' algorithmic bytes (9 B per wideband sample each) / wall time per step; per direction: "
                   "HIP events on its own stream around the K timed steps / K (the two spans overlap in time)",
            "rx": direction(rx_ms, rxc["kernel"]), "tx": direction(tx_ms, txc["kernel"]),
            "alone_kernel_ms": {k: round(v, 4) for k, v in alone.items()},
            "sum_alone_ms": round(alone["rx"] + alone["tx"], 4),
            # what running the two directions side by side buys over running them one after the other: the step's wall
            # time over the sum of the two kernels alone.  ~1.0 (measured 1.016 in round 4): both kernels sit at the
            # package power cap alone, so two streams share one power budget -- "full duplex" here is concurrency of the
            # API (read and write in flight together, SoapySX.cpp:878 / :979), not extra throughput
            "duplex_vs_serial": round(ms_per_step / (alone["rx"] + alone["tx"]), 4),
            "fp32_TFLOPs": round(128.0 * 2 * n / (ms_per_step * 1e-3) / 1e12, 2),
            "board": board,
        },
    }
    for k in ("power_w", "power_cap_w", "gfx_mhz_smi"):
        line["roofline"][k] = board.get(k)
    if not args.no_through_device:
        try:
            line["timed_loop"] = duplex_timed_loop()
            verified = verified and bool(line["timed_loop"].get("latency_check_passed"))
            line["verified"] = verified
        except Exception as e:
            line["timed_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["verified"] = verified = False
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(rxc)
        line["cpu_baseline"]["note_config3"] = "the RX direction's filter (256-tap decimate-by-8) on the host cores"
    print(json.dumps(line), flush=True)
    if not verified:
        sys.exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip's power management needs ~20 launches (~10 ms) of this kernel to settle
    # (kernel time swings 0.50 -> 0.77 -> 0.60 ms before it does), so warm up past that
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--log2-samples", type=int, default=28, help="wideband-side samples per GPU (log2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-through-device", action="store_true")
    ap.add_argument("--no-rate-table", action="store_true", help="skip the table of the reference's other rates (config 2 only)")
    ap.add_argument("--gather", default="torch", choices=["torch", "capi"],
                    help="N > 1: the gather of the decimated output through torch.distributed (nccl = RCCL) or through the "
                         "C ABI (sxfir_comm_gather over librccl directly)")
    ap.add_argument("--alone-seconds", type=float, default=0.6,
                    help="N > 1: seconds each rank times its kernel alone (and then all together) for the in-job efficiency figure")
    ap.add_argument("--asymmetric-taps", action="store_true",
                    help="--config 2 with a non-symmetric 128-tap filter: times decim4_tile_kernel<128>, the kernel any other "
                         "128-tap /4 plan runs (a row of its own, not the headline)")
    ap.add_argument("--settle", type=int, default=150,
                    help="untimed launches before the warm-up steps (clock settling)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if cfg["mode"] == "duplex":
        return main_duplex(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher.  Nothing below this line has run in this process, so no GPU call has
        # been made here; the ranks are fresh processes.
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import sxxcvr_amd
    from sxxcvr_amd import dist as sxdist
    from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, ClockProbe, StreamTimer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # SXFIR_DIST_BACKEND=gloo is a dry-run aid: it lets N ranks share the GPUs that exist (rank -> GPU
    # local_rank % device_count) and moves the collectives to the host, to exercise this code path on a
    # 1-GPU box.  The real multi-GPU run uses the default: nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("SXFIR_DIST_BACKEND")
    host_collectives = backend == "gloo"
    rank, local_rank, world = sxdist.env_rank()
    if not host_collectives and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d needs GPU %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    gpu_index = local_rank % torch.cuda.device_count() if host_collectives else local_rank
    torch.cuda.set_device(gpu_index)
    rank, local_rank, world = sxdist.init_process_group(backend=backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and args.config != "2":
        raise SystemExit("the multi-GPU layout (BASELINE config 4) runs --config 2's filter")
    if world > 1 and args.gather == "capi" and host_collectives:
        raise SystemExit("--gather capi is RCCL itself (sxfir_comm_* over librccl): it needs one GPU per rank, not the gloo stand-in")
    dev = torch.device("cuda", gpu_index)
    cdev = torch.device("cpu") if host_collectives else dev
    backend_name = dist.get_backend() if world > 1 else None
    # Who is in the job, as the job itself reports it: the communicator's size (not the environment's WORLD_SIZE) and the
    # PCI address of every rank's GPU, exchanged over a host-side control group (gloo).  The control group also carries
    # the turn-taking barriers of the alone / together kernel timing and the checksum tables of the gather checks, so that
    # the data-path communicator (nccl = RCCL) carries nothing but the data -- and a rank that waits its turn leaves its
    # GPU idle (a barrier over RCCL would park a spinning kernel on it).
    ident = gpu_identity(gpu_index)
    ident["rank"] = rank
    ctl, identities = None, [ident]
    if world > 1:
        ctl, ctl_name = dist.group.WORLD, backend_name
        if not host_collectives:
            gloo, why = None, ""
            try:
                gloo = dist.new_group(backend="gloo")
                dist.barrier(group=gloo)
            except Exception as e:
                gloo, why = None, "%s: %s" % (type(e).__name__, str(e)[:120])
            # every rank decides the SAME way: a rank whose gloo group came up while another's did not would otherwise wait
            # on a group its peer never joins (until the watchdog).  One all-reduce (min) of "mine works" over the data
            # communicator, which exists on every rank.
            ok = torch.tensor([1 if gloo is not None else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                ctl, ctl_name = gloo, "gloo"
            else:
                # no host-side group on this node (gloo could not connect the ranks): the control traffic rides the data
                # communicator instead -- waiting ranks then park RCCL's barrier kernel on their GPUs, which the line says
                ctl, ctl_name = dist.group.WORLD, "nccl (no gloo group on every rank%s)" % ((": " + why) if why else "")
        identities = [None] * world
        dist.all_gather_object(identities, ident, group=ctl)
        comm_ranks = dist.get_world_size()
        if comm_ranks != world:
            raise SystemExit("the communicator holds %d ranks, WORLD_SIZE says %d" % (comm_ranks, world))
        distinct = len({i.get("pci_bus_id") for i in identities})
        if not host_collectives and distinct != world:
            raise SystemExit("%d ranks but %d distinct GPUs (%s): RCCL needs one GPU per rank"
                             % (world, distinct, [i.get("pci_bus_id") for i in identities]))

    ratio, ntaps, decim = cfg["ratio"], cfg["ntaps"], cfg["mode"] == "decim"
    wide_per_gpu = 1 << args.log2_samples
    if ratio & (ratio - 1):
        wide_per_gpu -= wide_per_gpu % (512 * ratio)         # ratios 48, 96: whole tiles, and a block that ends on an output
    if world == 1:
        nchan_local, total_channels = 1, 1
        workload = "1xMI355X: " + cfg["name"]
    else:
        nchan_local, total_channels = 8, 8 * world
        workload = ("%dxMI355X: %d independent CF32 channels sharded 8/GPU, 128-tap decim-by-4 "
                    "(BASELINE config 4 layout)" % (world, total_channels))
    wide = wide_per_gpu // nchan_local                       # wideband samples per channel
    n_in = wide if decim else wide // ratio
    n_out = wide // ratio if decim else wide
    lo, hi = sxdist.shard_channels(total_channels, world, rank)
    assert hi - lo == nchan_local

    taps = sxxcvr_amd.design_lowpass(ntaps, ratio, 8.0, cfg["gain"])
    if args.asymmetric_taps:
        # any 128-tap /4 filter that is not bit-symmetric (a non-linear-phase design, an equaliser folded in) cannot
        # share taps between an output's two halves: such a plan runs decim4_tile_kernel<128> (taps in VGPRs) instead of
        # the headline kernel.  Here: the same low-pass with a 0.1 % slope across the taps.  Timed, verified against the
        # oracle with the same taps, reported as its own row -- never the headline.
        if args.config != "2":
            raise SystemExit("--asymmetric-taps is a variation of --config 2")
        taps = (taps.astype(np.float64) * (1.0 + 1e-3 * np.arange(ntaps) / ntaps)).astype(np.float32)
        cfg = dict(cfg, taps=taps, kernel="sxfir::decim4_tile_kernel<128> (taps in VGPRs: any 128-tap /4 filter)",
                   name=cfg["name"] + ", NON-symmetric taps")
        workload = workload.replace("(BASELINE config 2)", "(BASELINE config 2's shape, NON-symmetric taps: not the headline)")
    plan = sxxcvr_amd.Resampler(DECIMATE if decim else INTERPOLATE, taps, ratio, nchan=nchan_local, fmt=cfg["fmt"],
                                device=gpu_index)
    dt = torch.complex64 if cfg["fmt"] == "CF32" else torch.int32
    x = torch.empty((nchan_local, n_in), dtype=dt, device=dev)
    sxxcvr_amd.synth_fill(x, SEED, first_channel=lo, start=0, fmt=cfg["fmt"])
    y = torch.empty((nchan_local, n_out), dtype=dt, device=dev)
    torch.cuda.synchronize()
    xs = x.stride(0) if nchan_local > 1 else n_in
    ys = y.stride(0) if nchan_local > 1 else n_out
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        # one streaming pass: the block is the next 2^28 samples of a continuous stream (the filter
        # history carries over from the previous step, as it does in readStream)
        plan.process(x, out=y)

    # Setup, untimed and independent of --warmup: let the chip's power management settle on this kernel (its
    # time swings 0.49 -> 0.84 -> 0.60 ms over the first ~20 launches, LABBOOK.md section 7); same launches
    # as a step, then the stream restarts at position 0.
    # The first 20 launches on a chip that has been idle (what a bursty readStream caller sees, the pattern of
    # example/linear_repeater.py:57-69): reported as roofline.kernel_ms_first_20, never part of `value`.
    time.sleep(0.5)
    kernel_ms_first_20 = plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, 20, stream)
    plan.reset()
    torch.cuda.synchronize()
    if args.settle > 0:
        plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, args.settle, stream)
        plan.reset()
        torch.cuda.synchronize()

    # N > 1: per-GPU efficiency measured INSIDE this job (a line from another box differs by the 4-6 % box spread):
    # every rank times its kernel alone while the others idle (turns behind host-side barriers), then all ranks time it
    # together.  Node-level power / thermal coupling is the only thing that can cost per-GPU efficiency on a path with
    # no data-path collective, and this is what shows it.  Before any data-path collective; never part of `value`.
    per_rank = None
    if world > 1:
        sampler = BoardSampler(gpu_index, 0.02, bdf=ident.get("pci_bus_id"))
        sampler.start()

        def timed_for(seconds):
            t_a, ms = time.perf_counter(), []
            plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, 20, stream)       # out of the idle state
            t_b = time.perf_counter()
            while time.perf_counter() - t_b < seconds or not ms:
                ms.append(plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, 50, stream))
            return sum(ms) / len(ms), (t_b, time.perf_counter()), 50 * len(ms)

        mine = {}
        for turn in range(world):
            dist.barrier(group=ctl)
            if turn == rank:
                mine["alone_ms"], w_alone, mine["alone_launches"] = timed_for(args.alone_seconds)
        dist.barrier(group=ctl)
        mine["together_ms"], w_tog, mine["together_launches"] = timed_for(args.alone_seconds)
        dist.barrier(group=ctl)
        sampler.stop()                                   # (the windows below read what it recorded)
        for key, win in (("alone", w_alone), ("together", w_tog)):
            st = sampler.window(*win)
            mine["power_w_" + key] = st["power_w"] if st else None
            mine["gfx_mhz_" + key] = st["gfx_mhz_smi"] if st else None
        mine.update(rank=rank, gpu=ident.get("pci_bus_id"), gpu_index=gpu_index)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=ctl)
        plan.reset()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # the timed region: wall clock for `value`; HIP events on the launch stream around the same K launches for
    # the kernel's average duration (one launch per step, so the event span / K is that average plus whatever
    # gap the host leaves between launches)
    timer = StreamTimer(stream)
    t0 = time.perf_counter()
    timer.start()
    for _ in range(args.steps):
        step()
    timer.stop()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = timer.elapsed_ms() / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own kernel time over the same K timed steps (HIP events on its stream): the roofline fraction per GPU
        timed = [None] * world
        dist.all_gather_object(timed, float(kernel_ms), group=ctl)
        for r, ms in zip(per_rank, timed):
            r["timed_kernel_ms"] = round(ms, 6)
            r["timed_frac"] = round(cfg["bytes"] * wide_per_gpu / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    # the last step's output against the oracle (steady-state streaming: the block entered with its own tail
    # as history, unless it was the very first block of the stream)
    orc = load_oracle()
    first_block = (args.warmup + args.steps) <= 1
    ok1, cmp1 = verify(cfg, plan, x, y, n_in, lo, not first_block, orc)

    # the same launches once more, back to back from one C loop ...
    plan.reset()
    torch.cuda.synchronize()
    iters = min(max(args.steps, 40), 200)
    kernel_ms_loop = plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, iters, stream)
    # ... for about a second more with the board's telemetry sampled beside it (socket power, its cap, the GFX
    # clock the SMU reports: >= 10 samples) ...
    board, kernel_ms_telemetry = {"error": "not sampled (rank > 0)"}, None
    if rank == 0:
        sampler = BoardSampler(gpu_index, 0.02)
        sampler.start()
        t_tel, tel_ms = time.perf_counter(), []
        while time.perf_counter() - t_tel < 1.0:
            tel_ms.append(plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, 100, stream))
        board = sampler.stop()
        kernel_ms_telemetry = sum(tel_ms) / len(tel_ms)
    # ... and once more with the in-kernel shader clock read beside it (a few probe waves on a second stream;
    # their presence costs the kernel a few per cent, so this pass only supplies the clock)
    probe = ClockProbe(gpu_index, 8000)
    kernel_ms_probed = plan.time_passes_ptr(x.data_ptr(), n_in, xs, y.data_ptr(), ys, iters, stream)
    shader_mhz = probe.read()
    ok2, cmp2 = verify(cfg, plan, x, y, n_in, lo, False, orc)          # these passes started from zero history
    verified = ok1 and ok2
    if world > 1:
        t = torch.tensor([1.0 if verified else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        verified = bool(t.item() > 0.5)
    achieved = cfg["bytes"] * wide_per_gpu / (kernel_ms * 1e-3) / 1e9

    def frac_of(ms):
        return cfg["bytes"] * wide_per_gpu / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    tflops = cfg["flop"] * wide_per_gpu / (kernel_ms * 1e-3) / 1e12

    def emit(gather):
        """Rank 0 prints the one JSON line."""
        if rank != 0:
            return
        ms_per_step = elapsed / args.steps * 1e3
        value = world * wide_per_gpu * args.steps / elapsed / 1e6
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1 and args.log2_samples == 28 and not args.asymmetric_taps:
            try:
                tj = json.load(open(tp))
                ent = tj.get("configs", {}).get(args.config)
                if ent:
                    traffic = ent.get("hbm_bytes_per_launch")
                    traffic_source = "not measured in this run: rocprofv3 PMC passes FETCH_SIZE/WRITE_SIZE of the same " \
                                     "launch, " + ent.get("source", "profiles/")
            except Exception:
                traffic = None
        side = "input" if decim else "output"
        line = {
            "metric": "complex MS/s, %d-tap %s-by-%d %s (%s rate, whole job)" % (
                ntaps, "decim" if decim else "interp", ratio, cfg["fmt"], side),
            "value": round(value, 1),
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "verified": verified,
            "config": {
                "workload": workload,
                "bench_config": args.config + ("-asymmetric-taps" if args.asymmetric_taps else ""),
                "ntaps": ntaps, ("decim" if decim else "interp"): ratio, "format": cfg["fmt"],
                "channels_per_gpu": nchan_local, "wideband_samples_per_gpu": wide_per_gpu,
                "untimed_settle_launches": args.settle,
                "narrowband_MS/s": round(value / ratio, 1),
                "per_gpu_MS/s": round(value / world, 1),
                "verified_outputs": cmp1 + cmp2,
                "rccl_ranks": dist.get_world_size() if world > 1 else None,     # the communicator's own count
                "backend": backend_name,
                "control_group": ctl_name if world > 1 else None,
                "gpus": [i.get("pci_bus_id") for i in identities],
                "distinct_gpus": len({i.get("pci_bus_id") for i in identities}),
                "gpu_name": ident.get("name"), "gpu_arch": ident.get("arch"),
                # the host driver of this pool supports dmabuf IPC only: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's (and
                # torch's) cross-process buffer sharing fails with "hipIpcGetMemHandle: invalid argument"; the image exports
                # it, spawn_ranks sets it for its children if it is missing, and the line records what the ranks ran with
                "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": cfg["kernel"],
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": int(cfg["bytes"] * wide_per_gpu),
                "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
                "fp32_TFLOPs": round(tflops, 2),
                "shader_mhz": round(shader_mhz, 0),
                "kernel_ms_how": "HIP events on the launch stream around the K timed steps (one launch per step) / K",
                "kernel_ms_back_to_back_loop": round(kernel_ms_loop, 4),
                "kernel_ms_first_20": round(kernel_ms_first_20, 4),
                # the four regimes of the same kernel on this box, as fractions of the 8 TB/s peak, read off one record:
                # `frac` (the K timed steps, after the settle launches and the warm-up), the first 20 launches after 0.5 s
                # of idle (what a bursty readStream caller sees), the back-to-back C loop right after the host-side oracle
                # check (GPU idle for that long), and the one-second loop beside the telemetry sampling
                "frac_cold_first_20": round(frac_of(kernel_ms_first_20), 4),
                "frac_back_to_back_after_idle": round(frac_of(kernel_ms_loop), 4),
                "frac_while_sampled": round(frac_of(kernel_ms_telemetry), 4) if kernel_ms_telemetry else None,
                "kernel_ms_first_20_how": "the first 20 launches after 0.5 s of idle, before the settle launches: the "
                                          "clock transient a bursty caller sees; not part of value",
                "board": dict(board, kernel_ms_while_sampled=(round(kernel_ms_telemetry, 4)
                                                              if kernel_ms_telemetry else None)),
                "kernel_ms_beside_clock_probe": round(kernel_ms_probed, 4),
                "valu": {"achieved_TFLOPs": round(tflops, 2), "peak_TFLOPs_at_2400MHz": VALU_PEAK_TFLOPS,
                         "frac": round(tflops / VALU_PEAK_TFLOPS, 4),
                         "frac_at_measured_clock": round(tflops / (VALU_PEAK_TFLOPS * shader_mhz / 2400.0), 4)},
            },
        }
        for k in ("power_w", "power_cap_w", "gfx_mhz_smi"):      # the board's own figures beside shader_mhz
            line["roofline"][k] = board.get(k)
        if args.config == "5h":
            line["roofline"]["note"] = ("CF16 storage halves the bytes but not the 128 flop per sample: this leg is "
                                        "bound by fp32 VALU throughput at the clock the power management allows; "
                                        "see roofline.valu.  Since round 5 the half -> float conversion is made by the texture "
                                        "path on the way into LDS (buffer_load_format_x ... lds), not by the VALU")
        if per_rank is not None:
            # in-job efficiency: every rank's kernel alone (the others idle) against all ranks at once
            ratios = sorted(r["alone_ms"] / r["together_ms"] for r in per_rank)
            mean_alone = sum(r["alone_ms"] for r in per_rank) / world
            mean_tog = sum(r["together_ms"] for r in per_rank) / world
            for r in per_rank:
                for k in ("alone_ms", "together_ms"):
                    r[k] = round(r[k], 6)
            line["per_rank"] = per_rank
            fr = sorted(r["timed_frac"] for r in per_rank)
            line["roofline"]["frac_per_gpu"] = {"min": fr[0], "median": fr[len(fr) // 2], "max": fr[-1],
                                                "how": "every rank's own HIP-event time over the K timed steps (per_rank[].timed_kernel_ms); "
                                                       "roofline.frac is rank 0's"}
            line["efficiency_kernel_only"] = round(mean_alone / mean_tog, 4)
            line["efficiency_per_rank"] = {"min": round(ratios[0], 4), "median": round(ratios[len(ratios) // 2], 4),
                                           "max": round(ratios[-1], 4),
                                           "how": "alone_ms / together_ms of each rank's kernel (same launches, same buffers): "
                                                  "alone = the other ranks' GPUs idle, together = all ranks at once; "
                                                  "efficiency_kernel_only = mean(alone) / mean(together)"}
        if gather is not None:
            line["gather"] = gather
            checks = [c for c in (gather.get("check"), (gather.get("overlapped") or {}).get("check")) if c is not None]
            line["gather_verified"] = all(c["verified"] for c in checks) if checks else None
            if line["gather_verified"] is False:
                # a peer block that is not what its sender held: the line must not say verified, and the job fails
                line["verified"] = False
                gather_failed["v"] = True
        if world == 1 and not args.no_through_device:
            try:
                line["through_device"] = through_device()
            except Exception as e:                               # reported beside the value, never able to take it down
                line["through_device"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and args.config == "2" and not args.no_rate_table and not args.asymmetric_taps:
            try:
                line["size_curve"] = size_curve(dev)
            except Exception as e:
                line["size_curve"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                line["rate_table"] = rate_table(orc, dev, log2n=args.log2_samples)
            except Exception as e:
                line["rate_table"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline:
            # rank 0's host cores, at every N: a SCALE line at N > 1 is self-contained
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)

    # The exchange step (config 4's gather over xGMI) is reported beside `value`, never inside it, and must
    # not be able to take the line down with it: errors are recorded, and a watchdog prints the line without
    # the gather figures -- and ends the rank with a failure status -- if the collective does not come back.
    gather = None
    gather_failed = {"v": False}
    if world > 1:
        import threading
        finished, stage = threading.Event(), {"name": "serial", "gather": None}
        cert = GatherCertifier(cfg, plan, orc, world, rank, ctl, total_channels, nchan_local, n_in)
        stream_block = [0]

        def next_block():
            """The rank's input becomes the next block of its channels' synthetic stream and the filter restarts (zero
            history): what the following gather carries has never been gathered before.  Returns the block's first sample."""
            stream_block[0] += 1
            base = stream_block[0] * n_in
            sxxcvr_amd.synth_fill(x, SEED, first_channel=lo, start=base, fmt=cfg["fmt"])
            plan.reset()
            return base

        def step_into(out):
            plan.process(x, out=out)

        def watchdog():
            if finished.wait(240.0):
                return
            if stage["name"] == "serial":
                emit({"error": "gather did not complete within 240 s"})
                os._exit(3)
            # the plain gather is on record; only the pipelined form hung: report that and leave (a collective
            # that hangs cannot be cancelled)
            emit(dict(stage["gather"], overlapped={"error": "pipelined gather did not complete within 240 s"}))
            os._exit(5)                                 # a rank that abandons a live collective never reports success

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if args.gather == "capi":
                gather = measure_gather_capi(world, rank, gpu_index, y, total_channels, cdev, wide_per_gpu, elapsed, args.steps,
                                             step_into, cert, next_block)
            else:
                gather = measure_gather(world, y, total_channels, host_collectives, cdev, wide_per_gpu, elapsed, args.steps,
                                        cert, next_block, step_into)
        except Exception as e:
            gather = {"error": "%s: %s" % (type(e).__name__, e)}
        finished.set()
        if "error" not in gather and args.gather != "capi":
            finished, stage["name"], stage["gather"] = threading.Event(), "overlapped", gather
            threading.Thread(target=watchdog, daemon=True).start()
            try:
                gather.update(measure_gather_pipelined(world, y, total_channels, cdev, wide_per_gpu, step_into, cert, next_block))
            except Exception as e:
                gather["overlapped"] = {"error": "%s: %s" % (type(e).__name__, e)}
            finished.set()
    emit(gather)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not verified or gather_failed["v"]:
        sys.exit(4)


if __name__ == "__main__":
    main()
