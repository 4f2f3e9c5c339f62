"""TX to S32_LE wire words (Device arg wire=s32) at every ratio: the pass kernels (shipped) against interp_tile_kernel (SXFIR_IPASS=0),
profiling library; ms per 2^28 output words-pairs.   python tools/s32tx_ab.py"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, sxxcvr_amd
    from sxxcvr_amd.resampler import INTERPOLATE
    for ratio in (4, 8, 16, 32, 48, 96):
        wide = (1 << 28) // (512 * ratio) * (512 * ratio)
        taps = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, float(ratio))
        p = sxxcvr_amd.Resampler(INTERPOLATE, taps, ratio, fmt="S32", profiling=True)
        x = torch.empty(wide // ratio, dtype=torch.complex64, device="cuda")
        sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
        x *= 0.5
        y = torch.empty((wide, 2), dtype=torch.int32, device="cuda")
        for _ in range(30): p.process(x, out=y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): p.process(x, out=y)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40
        print("%s  x%-2d S32 out  %.3f ms per 2^28 outputs" % (sys.argv[1], ratio, t * 1e3 * (1 << 28) / wide), flush=True)
else:
    for rep in range(2):
        for name, env in (("pass kernels", {}), ("tile kernels", {"SXFIR_IPASS": "0"})):
            subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
