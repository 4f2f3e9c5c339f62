"""Run one decimator shape a few times (for rocprofv3 --pmc passes): python3 tools/onekernel.py D [fmt] [log2n]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE
D = int(sys.argv[1]); fmt = sys.argv[2] if len(sys.argv) > 2 else "CF32"; log2n = int(sys.argv[3]) if len(sys.argv) > 3 else 26
n = 1 << log2n
dt = torch.complex64 if fmt == "CF32" else torch.int32
x = torch.empty(n, dtype=dt, device="cuda"); sxxcvr_amd.synth_fill(x, 0x51255, 0, 0, fmt=fmt)
y = torch.empty(n // D, dtype=dt, device="cuda")
# SXFIR_PROF=1: the profiling build, whose SXFIR_* knobs (SXFIR_TILE_VARIANT, SXFIR_DENSE_NT ...) pick the kernel variant
p = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(32 * D, D), D, fmt=fmt, profiling=os.environ.get("SXFIR_PROF") == "1")
for _ in range(5): p.process(x, out=y)
torch.cuda.synchronize()
