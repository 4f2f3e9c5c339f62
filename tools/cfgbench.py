"""Throughput of the other BASELINE shapes (parity-test configs, not bench lines): for LABBOOK.md section 7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_GENERIC
def run(name, mode, ntaps, ratio, fmt, log2n, kernel=None, gain=1.0):
    n = 1 << log2n
    taps = sxxcvr_amd.design_lowpass(ntaps, ratio, 8.0, gain)
    p = sxxcvr_amd.Resampler(mode, taps, ratio, fmt=fmt)
    if kernel is not None: p.set_kernel(kernel)
    dt_in = torch.complex64 if fmt == "CF32" else torch.int32
    x = torch.empty(n, dtype=dt_in, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0, fmt=fmt)
    n_out = n // ratio if mode == DECIMATE else n * ratio
    y = torch.empty(n_out, dtype=dt_in, device="cuda")
    # warm up past the clock transient of the first launches (LABBOOK.md section 7); the slow
    # generic kernels get fewer repetitions
    fast = kernel is None
    for _ in range(100 if fast else 3): p.process(x, out=y)
    torch.cuda.synchronize(); iters = 100 if fast else 10; t0 = time.perf_counter()
    for _ in range(iters): p.process(x, out=y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    b = 8 if fmt == "CF32" else 4
    byt = b * (n + n_out)
    print("%-44s %8.3f ms  in %7.1f GS/s  out %7.1f GS/s  %6.0f GB/s (%.3f of 8 TB/s)" % (name, dt * 1e3, n / dt / 1e9, n_out / dt / 1e9, byt / dt / 1e9, byt / dt / 8e12))
run("config 2: 128-tap /4 CF32 tiled", DECIMATE, 128, 4, "CF32", 28)
run("config 3 RX: 256-tap /8 CF32 tiled", DECIMATE, 256, 8, "CF32", 28)
run("config 3 TX: 256-tap x8 CF32 tiled", INTERPOLATE, 256, 8, "CF32", 25, gain=8.0)
run("config 5: 1024-tap /32 CF32 tiled", DECIMATE, 1024, 32, "CF32", 28)
run("config 5: 1024-tap /32 CF16", DECIMATE, 1024, 32, "CF16", 28)
run("config 5: 1024-tap /32 CF32 generic", DECIMATE, 1024, 32, "CF32", 26, kernel=KERNEL_GENERIC)
run("128-tap /4 CF32 generic", DECIMATE, 128, 4, "CF32", 26, kernel=KERNEL_GENERIC)
