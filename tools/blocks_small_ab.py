"""Round 6 experiment (profiling build): /16 and /32 CF32 through decim_blocks_kernel -- one / two sixteen-column blocks per row,
scalar taps, waves by column group, the ROTATED contract -- against decim_dense_kernel<16 / 32> (VGPR taps; what ships).
SXFIR_BLOCKS_SMALL=1 selects the block form.  Parity of the block form against the oracle under rotation 1, then kernel time per
2^28 samples, alternating, BS_ROUNDS times.
    python3 tools/blocks_small_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, KERNEL_TILED
import oracle_lib

orc = oracle_lib.Oracle()
ROUNDS = int(os.environ.get("BS_ROUNDS", "3"))


def plan_for(D, small):
    os.environ["SXFIR_BLOCKS_SMALL"] = str(int(small))          # 0: dense (ships), 1: block form, rotated contract, 2: block form, unrotated
    h = sxxcvr_amd.design_lowpass(32 * D, D)
    p = sxxcvr_amd.Resampler(DECIMATE, h, D, profiling=True)
    p.set_kernel(KERNEL_TILED)
    return h, p


for D, form in ((16, 1), (32, 1), (16, 2), (32, 2)):
    h, p = plan_for(D, form)
    g = p.geometry(D * 512 * 40)
    c = p.contract
    print("/%d block form: kernel %s, contract %s rot %d" % (D, g["kernel"], tuple(c), c.rot), flush=True)
    lens = [2, 512, 512 * 40 + 78, 66]
    x = orc.synth_iq(0x51255, 3, 0, D * sum(lens))
    xd = torch.from_numpy(x).cuda()
    outs, pos = [], 0
    for n in lens:
        outs.append(p.process(xd[D * pos:D * (pos + n)]).cpu().numpy()); pos += n
    got = np.concatenate(outs)
    ref = orc.decim_f32(h, D, x, 2, 4, rot=c.rot)
    bad = int(np.count_nonzero(got.view(np.uint64) != ref.view(np.uint64)))
    print("   parity against the oracle (rot %d): %d of %d outputs differ" % (c.rot, bad, got.size), flush=True)
    p.close()

n_wide = 1 << 28
x = torch.empty(n_wide, dtype=torch.complex64, device="cuda")
sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
for D in (16, 32):
    y = torch.empty(n_wide // D, dtype=torch.complex64, device="cuda")
    plans = {small: plan_for(D, small)[1] for small in (0, 1, 2)}
    st = torch.cuda.current_stream().cuda_stream
    for r in range(ROUNDS):
        for small in (0, 1, 2):
            p = plans[small]
            p.time_passes_ptr(x.data_ptr(), n_wide, n_wide, y.data_ptr(), n_wide // D, 30, st)
            ms = min(p.time_passes_ptr(x.data_ptr(), n_wide, n_wide, y.data_ptr(), n_wide // D, 60, st) for _ in range(2))
            print("/%d %-32s %.2f us per 2^28 samples  (%.3f of 8 TB/s)" % (D, ("decim_dense_kernel (ships)", "decim_blocks_kernel, rotated", "decim_blocks_kernel, unrotated")[small], ms * 1e3,
                                                                          (8 + 8 / D) * n_wide / (ms * 1e-3) / 8e12), flush=True)
