"""Do consecutive passes of the /4 kernel gain from overlapping (the tail of pass k with the ramp-up of pass k+1)?
Feasibility probe: two plans (own history each) on two HIP streams taking alternate passes over the same 2^28-sample
input, against one plan on one stream.  T(n) = 33 us + 0.459 ms * n / 2^28 on one stream says a launch carries 7 % of
fixed cost at 2^28 samples."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE

n = 1 << 28
h = sxxcvr_amd.design_lowpass(128, 4)
DISTINCT = os.environ.get("OP_DISTINCT", "0") == "1"     # every stream reads its own copy of the input (no cache sharing between passes)
x = torch.empty(n, dtype=torch.complex64, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
xs = [x] + [x.clone() if DISTINCT else x for _ in range(7)]
ys = [torch.empty(n // 4, dtype=torch.complex64, device="cuda") for _ in range(8)]
plans = [sxxcvr_amd.Resampler(DECIMATE, h, 4) for _ in range(8)]
streams = [torch.cuda.Stream() for _ in range(8)]


def run(nstreams, steps):
    for i in range(steps):
        k = i % nstreams
        plans[k].process_ptr(xs[k].data_ptr(), n, n, ys[k].data_ptr(), n // 4, streams[k].cuda_stream)


def timed(nstreams, steps=60):
    run(nstreams, 150)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(nstreams, steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("distinct input buffers per stream:", DISTINCT)
for rep in range(2):
    for ns in (1, 2, 3, 4, 6, 8):
        ms = timed(ns)
        print("%d stream(s): %.4f ms per pass | %.0f GB/s algorithmic = %.3f of 8 TB/s" % (ns, ms, 10.0 * n / ms / 1e6, 10.0 * n / ms / 1e6 / 8000.0))
