"""Every selectable variant of the decimate-by-4 tile kernel (tools/kbench.py's A/B knobs) must
produce the bits of the default kernel: double-buffered LDS-DMA, contiguous-run schedule, packed
FMA arithmetic, SGPR-resident taps, different wave counts."""
import os

import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, KERNEL_TILED
from gpu_util import assert_bit_exact, to_cpu, to_gpu

pytestmark = pytest.mark.gpu

KNOBS = ("SXFIR_TILE_VARIANT", "SXFIR_OVERSUB", "SXFIR_OCC", "SXFIR_ABLATE", "SXFIR_SCHED")


@pytest.mark.parametrize("env", [
    {"SXFIR_TILE_VARIANT": "db"},
    {"SXFIR_TILE_VARIANT": "db", "SXFIR_SCHED": "1", "SXFIR_OVERSUB": "1"},
    {"SXFIR_SCHED": "1", "SXFIR_OVERSUB": "1"},
    {"SXFIR_OVERSUB": "3", "SXFIR_OCC": "5"},
    {"SXFIR_ABLATE": "3"},
    {"SXFIR_TILE_VARIANT": "sg"},
    {"SXFIR_TILE_VARIANT": "sg4"},
])
def test_variant_matches_oracle(oracle, monkeypatch, env):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = sxxcvr_amd.design_lowpass(128, 4)
    n = (1 << 20) + 4 * 77
    x = oracle.synth_iq(0x51255, 4, 0, n + 4096)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4)             # knobs are read at plan creation
    plan.set_kernel(KERNEL_TILED)
    y1 = to_cpu(plan.process(to_gpu(x[:n])))
    y2 = to_cpu(plan.process(to_gpu(x[n:])))                # exercises the fused history carry-over
    ref = oracle.decim_f32(h, 4, x, 2, 4)
    assert_bit_exact(np.concatenate([y1, y2]), ref, "variant %r" % env)
