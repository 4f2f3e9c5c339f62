"""Where does the power go?  Socket power and clocks (bench.py's BoardSampler) while a decimator runs back to back in
three forms of the profiling build: the whole kernel, its memory side alone (staging + stores, no FIR: SXFIR_ABLATE=1)
and its arithmetic alone (FIR out of LDS without staging: SXFIR_ABLATE=2).

    python3 tools/power_split.py [D=4] [seconds=2.5]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, ClockProbe
from bench import BoardSampler

D = int(sys.argv[1]) if len(sys.argv) > 1 else 4
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
n = 1 << 28
x = torch.empty(n, dtype=torch.complex64, device="cuda")
sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
y = torch.empty(n // D, dtype=torch.complex64, device="cuda")
taps = sxxcvr_amd.design_lowpass(32 * D, D)
st = torch.cuda.current_stream().cuda_stream
forms = [("whole kernel", 0), ("memory side alone", 1), ("arithmetic alone", 2)]
if D == 4:
    # PS_FORMS="label=variant:ablate,..." overrides (round 4: the shipped wide kernel and its memory side)
    os.environ["SXFIR_TILE_VARIANT"] = "t2:1:1088"
    forms = forms[:2] + [("arithmetic alone", 2)]
custom = os.environ.get("PS_FORMS")
if custom:
    forms = []
    for item in custom.split(","):
        label, spec = item.split("=")
        v, abl = spec.rsplit(":", 1)
        forms.append((label, (v, int(abl))))
for label, abl in forms:
    if isinstance(abl, tuple):
        os.environ["SXFIR_TILE_VARIANT"], abl = abl[0].replace(".", ":"), abl[1]
    os.environ["SXFIR_ABLATE"] = str(abl)
    try:
        plan = sxxcvr_amd.Resampler(DECIMATE, taps, D, profiling=True)
        for _ in range(3): plan.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // D, 50, st)
    except Exception as e:
        print("/%d %-18s not available: %s" % (D, label, e)); continue
    smp = BoardSampler(period_s=0.02); smp.start()
    ms, t0 = [], time.time()
    while time.time() - t0 < secs:
        ms.append(plan.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // D, 50, st))
    tel = smp.stop()
    print("/%d %-18s kernel ms med %.4f | power %s W (min %s max %s, cap %s) | SMU clock %s MHz" % (
        D, label, float(np.median(ms)), tel.get("power_w"), tel.get("power_w_min"), tel.get("power_w_max"),
        tel.get("power_cap_w"), tel.get("gfx_mhz_smi")))
    time.sleep(1.0)
