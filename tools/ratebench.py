"""Every sample rate the reference lists (SoapySX.cpp:180-208: master clock / {64, 128, 256, 512, 768, 1536}) as the device
runs it with decim=auto / interp=auto: ratio = divider / 16, 32 taps per phase.  Kernel time per 2^28 wideband samples, RX and TX.
   python tools/ratebench.py [fmt]        (fmt: CF32 (default) or CF16)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE

fmt = sys.argv[1] if len(sys.argv) > 1 else "CF32"
LOG2N = int(os.environ.get("RB_LOG2N", "28"))


def run(mode, ratio):
    n_wide = (1 << LOG2N) // (512 * ratio) * (512 * ratio)       # whole tiles of every kernel
    n_wide += int(os.environ.get("RB_ADD_TILES", "0")) * 512 * ratio      # ... plus a partial round of them (schedule tails)
    taps = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, 1.0 if mode == DECIMATE else float(ratio))
    p = sxxcvr_amd.Resampler(mode, taps, ratio, fmt=fmt, profiling=bool(os.environ.get("RB_PROF")))
    dt = {"CF32": torch.complex64, "CF16": torch.int32}[fmt]        # CF16: one 32-bit word (two halves) per sample
    per = 1
    n_in = n_wide if mode == DECIMATE else n_wide // ratio
    n_out = n_wide // ratio if mode == DECIMATE else n_wide
    x = torch.empty(n_in * per, dtype=dt, device="cuda")
    sxxcvr_amd.synth_fill(x, 0x51255, 0, 0, fmt=fmt)
    y = torch.empty(n_out * per, dtype=dt, device="cuda")
    p.process(x, out=y); torch.cuda.synchronize()
    t0 = time.perf_counter(); p.process(x, out=y); torch.cuda.synchronize(); one = time.perf_counter() - t0
    iters = max(3, min(100, int(0.5 / one)))
    for _ in range(iters): p.process(x, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): p.process(x, out=y)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    b = 8 if fmt != "CF16" else 4
    byt = b * (n_in + n_out)
    print("%s %-3s ratio %3d  %4d taps %9.3f ms per 2^%d wideband samples  %8.1f GS/s  %.3f of 8 TB/s" %
          ("RX /" if mode == DECIMATE else "TX x", fmt, ratio, 32 * ratio, t * 1e3 * (1 << LOG2N) / n_wide, LOG2N,
           n_wide / t / 1e9, byt / t / 8e12), flush=True)


RATIOS = [int(v) for v in os.environ.get("RB_RATIOS", "4,8,16,32,48,96").split(",")]
MODES = {"rx": (DECIMATE,), "tx": (INTERPOLATE,)}.get(os.environ.get("RB_MODE", ""), (DECIMATE, INTERPOLATE))
for mode in MODES:
    for ratio in RATIOS:
        try:
            run(mode, ratio)
        except Exception as e:
            print("ratio", ratio, "failed:", e, flush=True)
