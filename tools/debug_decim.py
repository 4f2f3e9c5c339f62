import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import sxxcvr_amd, oracle_lib
from sxxcvr_amd.resampler import DECIMATE, KERNEL_TILED, KERNEL_GENERIC
orc = oracle_lib.Oracle()
taps = sxxcvr_amd.design_lowpass(128, 4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
xs = orc.synth_iq(0x51255, 0, 0, n)
x = torch.from_numpy(xs).cuda()
for kern in (KERNEL_GENERIC, KERNEL_TILED):
    plan = sxxcvr_amd.Resampler(DECIMATE, taps, 4)
    plan.set_kernel(kern)
    y = plan.process(x); torch.cuda.synchronize()
    got = y.cpu().numpy()
    ref = orc.decim_f32(taps, 4, xs, 2, 4)
    bad = np.nonzero(got.view(np.uint64) != ref.view(np.uint64))[0]
    print("kernel", kern, "mismatches", bad.size, "of", got.size)
    if bad.size:
        print(" first", bad[:40])
        print(" mod 8 histogram", np.bincount(bad % 8, minlength=8))
        print(" mod 256 min/max", (bad % 256).min(), (bad % 256).max())
        for i in bad[:6]:
            print("  m=%d got=%r ref=%r" % (i, got[i], ref[i]))
        err = np.abs(got - ref)
        print(" max abs err", err.max(), "at", err.argmax())
