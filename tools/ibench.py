"""Interpolator throughput probe (profiling aid): tiled vs generic, L = KB_L."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import INTERPOLATE, KERNEL_GENERIC, KERNEL_TILED
L = int(os.environ.get("KB_L", "8"))
n = (1 << int(os.environ.get("KB_LOG2N", "25")))
x = torch.empty(n, dtype=torch.complex64, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
y = torch.empty(n * L, dtype=torch.complex64, device="cuda")
taps = sxxcvr_amd.design_lowpass(32 * L, L, 8.0, float(L))
for name, k in (("tiled", KERNEL_TILED), ("generic", KERNEL_GENERIC)):
    p = sxxcvr_amd.Resampler(INTERPOLATE, taps, L); p.set_kernel(k)
    fast = k == KERNEL_TILED
    for _ in range(100 if fast else 3): p.process(x, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter(); iters = 100 if fast else 10
    for _ in range(iters): p.process(x, out=y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print("%-8s L=%d  %.4f ms  in %.1f GS/s  out %.1f GS/s  %.0f GB/s algorithmic (%.3f of 8 TB/s)" % (
        name, L, dt * 1e3, n / dt / 1e9, n * L / dt / 1e9, (8 + 8 * L) * n / dt / 1e9, (8 + 8 * L) * n / dt / 8e12))
