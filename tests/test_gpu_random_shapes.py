"""Seeded random sweep of shapes through the C ABI on the GPU: any tap count, ratio, block sizes, channel
count and strides must give the oracle's bits with the plan's own contract, whichever kernel the library
picks (LDS-tiled or generic), across several calls of one stream (history carry-over)."""
import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE
from gpu_util import assert_bit_exact, to_cpu, to_gpu

pytestmark = pytest.mark.gpu


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ratio = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 32]))
        tpp = int(rng.choice([1, 2, 7, 16, 32, 33]))
        ntaps = ratio * tpp if rng.random() < 0.7 else int(rng.integers(1, 300))
        nchan = int(rng.choice([1, 1, 2, 3]))
        blocks = [int(v) for v in rng.integers(0, 9000, size=int(rng.integers(1, 5)))]
        pad = int(rng.choice([0, 0, 1, 6]))
        out.append((ratio, ntaps, nchan, blocks, pad))
    return out


@pytest.mark.parametrize("ratio,ntaps,nchan,blocks,pad", _cases(20251, 40))
def test_random_decimator_shapes(oracle, ratio, ntaps, nchan, blocks, pad):
    import torch
    h = (np.random.default_rng(ntaps * 131 + ratio).standard_normal(ntaps) / max(ntaps, 1)).astype(np.float32)
    total = sum(blocks)
    xs = [oracle.synth_iq(0x51255, 40 + c, 0, total) for c in range(nchan)]
    plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio, nchan=nchan)
    js, cw = plan.contract
    got = [[] for _ in range(nchan)]
    pos = 0
    for n in blocks:
        # channel-major input with `pad` unused samples between channels (stride != length)
        buf = np.zeros((nchan, n + pad), dtype=np.complex64)
        for c in range(nchan):
            buf[c, :n] = xs[c][pos:pos + n]
        xg = to_gpu(buf)
        y = plan.process(xg[:, :n] if nchan > 1 else xg[0, :n])
        torch.cuda.synchronize()
        y = to_cpu(y).reshape(nchan, -1)
        for c in range(nchan):
            got[c].append(y[c])
        pos += n
    for c in range(nchan):
        ref = oracle.decim_f32(h, ratio, xs[c], js, cw, rot=plan.contract.rot)
        assert_bit_exact(np.concatenate(got[c]), ref, "decim ratio=%d ntaps=%d chan=%d" % (ratio, ntaps, c))


@pytest.mark.parametrize("ratio,ntaps,nchan,blocks,pad", _cases(777, 25))
def test_random_interpolator_shapes(oracle, ratio, ntaps, nchan, blocks, pad):
    import torch
    ntaps = max(ratio, (ntaps // ratio) * ratio)               # interpolators need whole phases
    blocks = [min(b, 1500) for b in blocks]
    h = (np.random.default_rng(ntaps * 17 + ratio).standard_normal(ntaps) / max(ntaps, 1)).astype(np.float32)
    total = sum(blocks)
    xs = [oracle.synth_iq(0x51255, 90 + c, 0, total) for c in range(nchan)]
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, ratio, nchan=nchan)
    groups = plan.contract[0]
    got = [[] for _ in range(nchan)]
    pos = 0
    for n in blocks:
        buf = np.zeros((nchan, n + pad), dtype=np.complex64)
        for c in range(nchan):
            buf[c, :n] = xs[c][pos:pos + n]
        xg = to_gpu(buf)
        y = plan.process(xg[:, :n] if nchan > 1 else xg[0, :n])
        torch.cuda.synchronize()
        y = to_cpu(y).reshape(nchan, -1)
        for c in range(nchan):
            got[c].append(y[c])
        pos += n
    for c in range(nchan):
        ref = oracle.interp_f32(h, ratio, xs[c], groups)
        assert_bit_exact(np.concatenate(got[c]), ref, "interp ratio=%d ntaps=%d chan=%d" % (ratio, ntaps, c))


@pytest.mark.parametrize("ratio,ntaps,nchan,blocks,pad", _cases(4242, 30))
@pytest.mark.parametrize("fmt", ["CF16", "S32"])
def test_random_decimator_shapes_other_formats(oracle, fmt, ratio, ntaps, nchan, blocks, pad):
    """The same sweep through the CF16 storage path (half in, fp32 arithmetic, half out rounded once) and the
    S32 I2S wire-word front end (convert_rx_buffer folded into the kernels, SX.cpp:103-112)."""
    import torch
    if nchan > 1:
        pad = pad & ~1                                            # keeps CF16 rows 8-byte aligned like the API asks
    h = (np.random.default_rng(ntaps * 31 + ratio).standard_normal(ntaps) / max(ntaps, 1)).astype(np.float32)
    total = sum(blocks)
    rng = np.random.default_rng(total + ratio)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio, nchan=nchan, fmt=fmt)
    js, cw = plan.contract
    if fmt == "CF16":
        raw = [oracle.f32_to_f16(oracle.synth_iq(0x51255, 60 + c, 0, total).view(np.float32)).view(np.uint32).view(np.int32)
               for c in range(nchan)]                             # one int32 word = half I, half Q
        xs = [oracle.f16_to_f32(r.view(np.uint16)).view(np.complex64) for r in raw]
    else:
        raw = [rng.integers(-2 ** 31, 2 ** 31, size=2 * total, dtype=np.int64).astype(np.int32).reshape(-1, 2)
               for _ in range(nchan)]
        xs = [oracle.convert_rx(r.ravel()) for r in raw]
    got = [[] for _ in range(nchan)]
    pos = 0
    for n in blocks:
        shape = (nchan, n + pad) if fmt == "CF16" else (nchan, n + pad, 2)
        buf = np.zeros(shape, dtype=np.int32)
        for c in range(nchan):
            buf[c, :n] = raw[c][pos:pos + n]
        xg = to_gpu(buf)
        y = plan.process(xg[:, :n] if nchan > 1 else xg[0, :n])
        torch.cuda.synchronize()
        y = to_cpu(y)
        y = y.reshape(nchan, -1)
        for c in range(nchan):
            got[c].append(y[c])
        pos += n
    for c in range(nchan):
        ref = oracle.decim_f32(h, ratio, xs[c], js, cw, rot=plan.contract.rot)
        if fmt == "CF16":
            want = oracle.f32_to_f16(ref.view(np.float32))
            assert np.array_equal(np.concatenate(got[c]).view(np.uint16), want), "CF16 ratio=%d ntaps=%d" % (ratio, ntaps)
        else:
            assert_bit_exact(np.concatenate(got[c]), ref, "S32 ratio=%d ntaps=%d chan=%d" % (ratio, ntaps, c))
