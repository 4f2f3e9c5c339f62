"""Soak of the two dealt forms (round 6): decim_blocks_kernel<..., SPLIT> hands block values from workgroup to workgroup through
HBM with device-scope stores / loads and one atomic add per item -- the form the microarchitecture guide measured, "not an
architectural guarantee" -- and interp8_pass_kernel<..., PBSPLIT> deals phase blocks.  Every launch's output is compared, bit for
bit on the GPU, with the first launch's (itself checked against the oracle on its first outputs): a stale or torn hand-off would
show as a differing word.
    python3 tools/soak_split.py [seconds per case]        (default 4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_TILED
import oracle_lib

orc = oracle_lib.Oracle()
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
total_launches = total_bad = 0
for mode, ratio, nchan, lg in [(DECIMATE, 96, 1, 18), (DECIMATE, 96, 1, 20), (DECIMATE, 96, 1, 22), (DECIMATE, 96, 1, 24), (DECIMATE, 96, 1, 26),
                               (DECIMATE, 48, 1, 20), (DECIMATE, 48, 1, 22), (DECIMATE, 48, 1, 25), (DECIMATE, 96, 4, 22), (DECIMATE, 48, 8, 23),
                               (INTERPOLATE, 96, 1, 20), (INTERPOLATE, 96, 1, 24), (INTERPOLATE, 48, 2, 22), (INTERPOLATE, 32, 1, 22)]:
    gain = 1.0 if mode == DECIMATE else float(ratio)
    h = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, gain)
    plan = sxxcvr_amd.Resampler(mode, h, ratio, nchan=nchan)
    plan.set_kernel(KERNEL_TILED)
    wide = ((1 << lg) // nchan) // (4 * ratio) * (4 * ratio)                     # per channel; aligned channel rows
    n_in = wide if mode == DECIMATE else wide // ratio
    n_out = wide // ratio if mode == DECIMATE else wide
    g = plan.geometry(n_in)
    x = torch.empty((nchan, n_in), dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
    y0 = torch.empty((nchan, n_out), dtype=torch.complex64, device="cuda")
    y = torch.empty_like(y0)
    plan.reset(); plan.process(x if nchan > 1 else x[0], out=y0 if nchan > 1 else y0[0]); torch.cuda.synchronize()
    # the first launch against the oracle (first 3000 outputs of channel 0)
    k = min(3000, n_out)
    xs = orc.synth_iq(0x51255, 0, 0, k * ratio if mode == DECIMATE else (k + ratio - 1) // ratio)
    ref = orc.decim_f32(h, ratio, xs, 2, 4, rot=plan.contract.rot)[:k] if mode == DECIMATE else orc.interp_f32(h, ratio, xs, 2)[:k]
    ok0 = np.array_equal(y0[0, :k].cpu().numpy().view(np.uint64), ref.view(np.uint64))
    ref_words = torch.view_as_real(y0).view(torch.int32)
    launches = bad = 0
    t0 = time.time()
    while time.time() - t0 < SECONDS:
        for _ in range(20):
            y.zero_()
            plan.reset(); plan.process(x if nchan > 1 else x[0], out=y if nchan > 1 else y[0])
            bad += int((torch.view_as_real(y).view(torch.int32) != ref_words).any().item())
            launches += 1
    print("%s%-3d %d ch 2^%d: %s, %d tiles x%d, %d workgroups on %d slots: first launch %s the oracle; %d launches, %d differ" % (
        "/" if mode == DECIMATE else "x", ratio, nchan, lg, g["kernel"], g["n_tiles"], g["split"], g["workgroups"], g["resident"],
        "equals" if ok0 else "DIFFERS FROM", launches, bad), flush=True)
    total_launches += launches; total_bad += bad + (0 if ok0 else 1)
    plan.close(); del x, y, y0
print("total: %d launches, %d bad" % (total_launches, total_bad))
sys.exit(1 if total_bad else 0)
