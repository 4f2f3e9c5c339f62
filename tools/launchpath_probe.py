"""Why are kernels launched by sxfir_time_* slower than the same kernels launched by sxfir_decimate from Python?
Times the same 2^28-sample pass through several launch paths, interleaved (profiling aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE
n = 1 << 28
x = torch.empty(n, dtype=torch.complex64, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
y = torch.empty(n // 4, dtype=torch.complex64, device="cuda")
p = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(128, 4), 4)
st = torch.cuda.current_stream().cuda_stream
p.time_passes_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 4, 150, st)
def timed(fn, k=50):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k, (time.perf_counter() - t0) * 1e3 / k
def steps(k):
    for _ in range(k): p.process(x, out=y)
def ptr(k):
    for _ in range(k): p.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 4, st)
def passes(k):
    p.time_passes_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 4, k, st)
def passes1(k):
    for _ in range(k): p.time_passes_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 4, 1, st)
def steps_sleep(k):
    for _ in range(k):
        p.process(x, out=y); time.sleep(0.0004)
for r in range(4):
    for name, fn in (("process() x50", steps), ("process_ptr() x50", ptr), ("time_passes(50)", passes), ("time_passes(1) x50", passes1),
                     ("process() + 0.4 ms host sleep x50", steps_sleep)):
        ev, wall = timed(fn)
        print("round %d  %-36s event %.4f ms/launch   wall %.4f ms/launch" % (r, name, ev, wall), flush=True)
