"""One pass of the /4 kernel split into S sub-passes over consecutive parts of the input, launched on S streams at
once and joined (every stream waits for all sub-passes of pass k before pass k+1 starts): does the gain of
tools/overlap_probe.py (several kernels resident at a time) survive a join per pass?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE

n = 1 << 28
h = sxxcvr_amd.design_lowpass(128, 4)
x = torch.empty(n, dtype=torch.complex64, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
y = torch.empty(n // 4, dtype=torch.complex64, device="cuda")
SMAX = 16
plans = [sxxcvr_amd.Resampler(DECIMATE, h, 4) for _ in range(SMAX)]
streams = [torch.cuda.Stream() for _ in range(SMAX)]


def run(S, steps):
    c = n // S
    for i in range(steps):
        evs = []
        for s in range(S):
            plans[s].process_ptr(x.data_ptr() + 8 * s * c, c, c, y.data_ptr() + 8 * s * (c // 4), c // 4, streams[s].cuda_stream)
            e = torch.cuda.Event()
            e.record(streams[s])
            evs.append(e)
        if S > 1:
            for s in range(S):
                for e in evs:
                    streams[s].wait_event(e)


def timed(S, steps=40):
    run(S, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(S, steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for S in (1, 2, 4, 8, 16):
        ms = timed(S)
        print("%2d sub-passes: %.4f ms per pass | %.0f GB/s algorithmic = %.3f of 8 TB/s" % (S, ms, 10.0 * n / ms / 1e6, 10.0 * n / ms / 1e6 / 8000.0))
