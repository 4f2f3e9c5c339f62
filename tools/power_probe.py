"""Board telemetry while a decimator runs back to back: socket power, power cap, GFX clock (amdsmi), the in-kernel
shader clock (sxfir_clock_probe) and the kernel time -- once on the bench's random IQ and once on an all-zero
input, the same binary.  If the kernel were bound by its own structure the two would take the same time; if the
package power cap sets the clock, zeros (no toggling in the FMA datapath) run faster.

    python3 tools/power_probe.py [D=4 | xL for the interpolator, e.g. x8] [seconds=3]
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, ClockProbe
from bench import BoardSampler

arg = sys.argv[1] if len(sys.argv) > 1 else "4"
interp = arg.startswith("x")
D = int(arg[1:]) if interp else int(arg)
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n = (1 << 28) // D if interp else 1 << 28          # input samples: 2^28 on the wideband side either way
n_out = n * D if interp else n // D
x = torch.empty(n, dtype=torch.complex64, device="cuda")
y = torch.empty(n_out, dtype=torch.complex64, device="cuda")
plan = sxxcvr_amd.Resampler(INTERPOLATE if interp else DECIMATE,
                            sxxcvr_amd.design_lowpass(32 * D, D, 8.0, float(D) if interp else 1.0), D)
st = torch.cuda.current_stream().cuda_stream

s0 = BoardSampler(period_s=0.05)
s0.start(); time.sleep(1.0); idle = s0.stop()
print("idle:", idle)
for label in ("random IQ", "all-zero IQ", "random IQ again"):
    if label.startswith("all-zero"): x.zero_()
    else: sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
    torch.cuda.synchronize()
    plan.reset()
    for _ in range(3): plan.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n_out, 50, st)   # settle
    smp = BoardSampler(period_s=0.02)
    smp.start()
    ms, t0 = [], time.time()
    while time.time() - t0 < secs:
        ms.append(plan.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n_out, 50, st))
    tel = smp.stop()
    probe = ClockProbe(duration_us=8000)
    ms2 = plan.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n_out, 40, st)
    mhz = probe.read()
    print(("x%d" if interp else "/%d") % D + " %-16s kernel ms med %.4f min %.4f | in-kernel shader clock %.0f MHz (beside the probe: %.4f ms) | %s" % (
        label, float(np.median(ms)), min(ms), mhz, ms2, tel))
