"""Size curve of every rate of the reference's table (SoapySX.cpp:180-208; ratio = divider / 16, 32 taps per phase), both
directions: kernel time per WIDEBAND sample at call sizes 2^16 .. 2^28 wideband samples, next to what each launch looks like
(tiles, workgroups, workgroup slots of the chip: sxfir_launch_geometry) -- readStream / writeStream never issue 2^28-sample
calls (the reference's blocks are 256 .. 8192 samples; the Device's chains batch them, see the last
lines of the output).

    python3 tools/sizebench.py [CF32|CF16|S32]        SB_LOG2=16,18,..  SB_RATIOS=4,8,..  SB_MODE=rx|tx  SB_PROF=1

Times are sxfir_time_decimate / sxfir_time_interpolate: HIP events on the launch stream around back-to-back launches from one
C loop (no Python per launch); below a few microseconds they are the launch rate of the runtime, not the kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE

fmt = sys.argv[1] if len(sys.argv) > 1 else "CF32"
LOG2 = [int(v) for v in os.environ.get("SB_LOG2", "16,18,20,22,24,26,28").split(",")]
RATIOS = [int(v) for v in os.environ.get("SB_RATIOS", "4,8,16,32,48,96").split(",")]
MODES = {"rx": (DECIMATE,), "tx": (INTERPOLATE,)}.get(os.environ.get("SB_MODE", ""), (DECIMATE, INTERPOLATE))
PROF = bool(os.environ.get("SB_PROF"))


def run(mode, ratio, rows):
    taps = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, 1.0 if mode == DECIMATE else float(ratio))
    p = sxxcvr_amd.Resampler(mode, taps, ratio, fmt=fmt, profiling=PROF)
    if fmt == "S32":
        dt_in = torch.int64 if mode == DECIMATE else torch.complex64        # 8 bytes per sample either side
        dt_out = torch.complex64 if mode == DECIMATE else torch.int64
    else:
        dt_in = dt_out = {"CF32": torch.complex64, "CF16": torch.int32}[fmt]
    for lg in LOG2:
        n_wide = (1 << lg) // ratio * ratio                  # a call of 2^lg wideband samples, whole outputs
        n_in = n_wide if mode == DECIMATE else n_wide // ratio
        n_out = n_wide // ratio if mode == DECIMATE else n_wide
        x = torch.empty(n_in, dtype=dt_in, device="cuda")
        if fmt == "S32":
            x.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1) if mode == DECIMATE else sxxcvr_amd.synth_fill(x, 0x51255, 0, 0)
        else:
            sxxcvr_amd.synth_fill(x, 0x51255, 0, 0, fmt=fmt)
        y = torch.empty(n_out, dtype=dt_out, device="cuda")
        g = p.geometry(n_in)
        st = torch.cuda.current_stream().cuda_stream
        ms = p.time_passes_ptr(x.data_ptr(), n_in, n_in, y.data_ptr(), n_out, 3, st)
        iters = max(5, min(2000, int(100.0 / max(ms, 1e-3))))                # ~0.1 s per point
        p.time_passes_ptr(x.data_ptr(), n_in, n_in, y.data_ptr(), n_out, iters, st)
        ms = min(p.time_passes_ptr(x.data_ptr(), n_in, n_in, y.data_ptr(), n_out, iters, st) for _ in range(3))
        rows[(mode, ratio, lg)] = ms * 1e6 / n_wide
        print("%s%-3d %-4s 2^%-2d  %9.2f us  %7.3f ps/sample  %7.1f GS/s wideband   %-22s tile %6d  tiles %7d x%d  workgroups %6d  slots %5d" % (
            "RX /" if mode == DECIMATE else "TX x", ratio, fmt, lg, ms * 1e3, ms * 1e9 / n_wide, n_wide / ms / 1e6,
            g["kernel"], g["tile_samples"], g["n_tiles"], g["split"], g["workgroups"], g["resident"]), flush=True)
        del x, y
    p.close()


rows = {}
for mode in MODES:
    for ratio in RATIOS:
        try:
            run(mode, ratio, rows)
        except Exception as e:
            print("ratio", ratio, "failed:", e, flush=True)

print("\nns per wideband sample relative to /32 (RX) and x32 (TX) at the same call size:")
for mode in MODES:
    for ratio in RATIOS:
        cells = []
        for lg in LOG2:
            a, b = rows.get((mode, ratio, lg)), rows.get((mode, 32, lg))
            cells.append("%5.2f" % (a / b) if a and b else "  -  ")
        print("%s%-3d  " % ("RX /" if mode == DECIMATE else "TX x", ratio) + "  ".join("2^%d %s" % (lg, c) for lg, c in zip(LOG2, cells)))

print("\nbatches the Device's chains launch (GpuChains.hpp: RxChain kMinBatch 4096 .. max_batch stream samples per pass, with"
      " 2 * max_batch * ratio * channels <= 2^27 wideband samples):")
for ratio in RATIOS:
    mb = 4096
    while mb < (1 << 20) and 2 * mb * ratio <= (1 << 27):
        mb *= 2
    print("  ratio %3d: %7d .. %7d stream samples = 2^%.1f .. 2^%.1f wideband samples per pass" % (
        ratio, 4096, mb, __import__("math").log2(4096 * ratio), __import__("math").log2(mb * ratio)))
