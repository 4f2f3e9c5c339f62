"""x8 interpolator A/B (profiling build): interp8_pass_kernel (scalar taps, four passes per tile; SXFIR_IPASS=1, the
product) against interp_tile_kernel (taps in VGPRs; SXFIR_IPASS=0), generations by SXFIR_OVERSUB; long interleaved
visits (KB_SETTLE untimed + KB_ITERS timed launches), random IQ and (KB_ZERO=1) all-zero input.
    python3 tools/ibench2.py pass:4 tile:4 pass:8 ...        spec = kernel:oversub"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import INTERPOLATE
L = 8
n = 1 << (int(os.environ.get("KB_LOG2N", "28")) - 3)
rounds, iters, settle = int(os.environ.get("KB_ROUNDS", "5")), int(os.environ.get("KB_ITERS", "200")), int(os.environ.get("KB_SETTLE", "100"))
x = torch.empty(n, dtype=torch.complex64, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
if os.environ.get("KB_ZERO") == "1":
    x.zero_(); print("# all-zero input")
y = torch.empty(n * L, dtype=torch.complex64, device="cuda")
taps = sxxcvr_amd.design_lowpass(32 * L, L, 8.0, float(L))
specs = sys.argv[1:] or ["pass:4", "tile:4"]
plans = []
for k, sp in enumerate(specs):
    kern, ov = sp.split(":")
    os.environ["SXFIR_IPASS"] = {"pass": "1", "pass4": "4"}.get(kern, "0")      # pass: the shipped form (two inputs per lane); pass4: four
    os.environ["SXFIR_OVERSUB"] = ov
    plans.append(sxxcvr_amd.Resampler(INTERPOLATE, taps, L, fmt=os.environ.get("KB_FMT", "CF32"), profiling=True))   # KB_FMT=S32: wire-word output (same 8 bytes per sample)
st = torch.cuda.current_stream().cuda_stream
res, ref = [[] for _ in specs], None
for r in range(rounds):
    for k, p in enumerate(plans):
        p.reset()
        p.time_passes_ptr(x.data_ptr(), n, n, y.data_ptr(), n * L, settle, st)
        res[k].append(p.time_passes_ptr(x.data_ptr(), n, n, y.data_ptr(), n * L, iters, st))
        if r == 0:
            torch.cuda.synchronize()
            chk = torch.view_as_real(y).view(torch.int32).sum(dtype=torch.int64).item()
            ref = chk if ref is None else ref
            print("spec", specs[k], "checksum", "same" if chk == ref else "DIFFERENT")
for k, sp in enumerate(specs):
    a = np.array(res[k]); gbs = 9.0 * n * L / (a * 1e-3) / 1e9
    print("%-10s ms med %.4f min %.4f max %.4f | GB/s med %.0f | frac of 8TB/s %.3f" % (sp, np.median(a), a.min(), a.max(), np.median(gbs), np.median(gbs) / 8000))
