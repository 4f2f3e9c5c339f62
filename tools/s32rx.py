"""RX from S32_LE wire words at every ratio of the reference's table (SoapySX.cpp:180-208): kernel time per 2^28 wideband
samples of the SXFIR_S32 decimator plans (int -> float on load fused, convert_rx_buffer SoapySX.cpp:103-112).
    python3 tools/s32rx.py        RB_RATIOS=4,8,..  RB_PROF=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE
for ratio in [int(v) for v in os.environ.get("RB_RATIOS", "4,8,16,32,48,96").split(",")]:
    wide = (1 << 28) // (512 * ratio) * (512 * ratio)
    taps = sxxcvr_amd.design_lowpass(32 * ratio, ratio)
    p = sxxcvr_amd.Resampler(DECIMATE, taps, ratio, fmt="S32", profiling=bool(os.environ.get("RB_PROF")))
    x = torch.randint(-2**31, 2**31 - 1, (wide, 2), dtype=torch.int32, device="cuda")
    y = torch.empty(wide // ratio, dtype=torch.complex64, device="cuda")
    for _ in range(30): p.process(x, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): p.process(x, out=y)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40
    print("RX /%-2d S32 words in  %.3f ms per 2^28 samples" % (ratio, t * 1e3 * (1 << 28) / wide), flush=True)
