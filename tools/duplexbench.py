"""BASELINE config 3 as a full-duplex load: the 256-tap decimate-by-8 RX kernel and the 256-tap interpolate-by-8 TX
kernel running at the same time on two HIP streams over 2^28 wideband samples each, against each of them alone.
(bench.py --config 3rx / 3tx are the two halves on their own; this is what they make of the GPU together.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE

LOG2N = int(os.environ.get("DB_LOG2N", "28"))
n = 1 << LOG2N
dev = torch.device("cuda", 0)
rx_plan = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(256, 8), 8)
tx_plan = sxxcvr_amd.Resampler(INTERPOLATE, sxxcvr_amd.design_lowpass(256, 8, 8.0, 8.0), 8)
xr = torch.empty(n, dtype=torch.complex64, device=dev); sxxcvr_amd.synth_fill(xr, 0x51255, 0, 0)
yr = torch.empty(n // 8, dtype=torch.complex64, device=dev)
xt = torch.empty(n // 8, dtype=torch.complex64, device=dev); sxxcvr_amd.synth_fill(xt, 0x51255, 1, 0)
yt = torch.empty(n, dtype=torch.complex64, device=dev)
s_rx, s_tx = torch.cuda.Stream(), torch.cuda.Stream()


def run(which, steps):
    for _ in range(steps):
        if "rx" in which:
            rx_plan.process_ptr(xr.data_ptr(), n, n, yr.data_ptr(), n // 8, s_rx.cuda_stream)
        if "tx" in which:
            tx_plan.process_ptr(xt.data_ptr(), n // 8, n // 8, yt.data_ptr(), n, s_tx.cuda_stream)


def timed(which, steps=60):
    run(which, 150)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(which, steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for which in ("rx", "tx", "rx+tx", "rx", "tx", "rx+tx"):
    ms = timed(which)
    wide = n * (2 if which == "rx+tx" else 1)
    print("%-6s %.4f ms per step | %.0f G wideband samples/s | %.0f GB/s algorithmic (9 B per wideband sample) = %.3f of 8 TB/s" % (
        which, ms, wide / ms / 1e6, 9.0 * wide / ms / 1e6, 9.0 * wide / ms / 1e6 / 8000.0))
