"""BASELINE.json configs 3, 4 and 5 at full size on the GPU: size-independent checks (no CPU pass over the
whole stream): spot windows against the oracle bit for bit (the source is counter based, any window can be
regenerated on the CPU), chunking invariance, linearity under exact scalings, multi-channel == per-channel."""
import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_TILED
from gpu_util import assert_bit_exact, to_cpu, to_gpu

pytestmark = pytest.mark.gpu

SEED = 0x51255
LOG2N = 28          # wide-rate samples per pass, as in bench.py


def _sync():
    import torch
    torch.cuda.synchronize()


@pytest.mark.parametrize("D", [8, 32])
def test_full_size_decimators(oracle, D):
    """Config 3 RX (256 taps, /8) and config 5 (1024 taps, /32), CF32, 2^28 input samples."""
    import torch
    nt = 32 * D
    h = sxxcvr_amd.design_lowpass(nt, D)
    n = 1 << LOG2N
    x = torch.empty(n, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 0, 0)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, D)
    plan.set_kernel(KERNEL_TILED)
    js, cw = plan.contract
    y = plan.process(x)
    _sync()
    total = n // D
    assert y.numel() == total
    # (1) windows: start, a workgroup-tile seam, middle, end.  The oracle is fed nt samples of history.
    for m0 in (0, 255, 128 * 4001 - 5, total // 2 + 3, total - 200):
        cnt = min(200, total - m0)
        got = to_cpu(y[m0:m0 + cnt])
        if m0 < 32:
            ref = oracle.decim_f32(h, D, oracle.synth_iq(SEED, 0, 0, D * (m0 + cnt)), js, cw, rot=plan.contract.rot)[m0:m0 + cnt]
        else:
            w = oracle.synth_iq(SEED, 0, D * m0 - nt, nt + D * cnt)
            ref = oracle.decim_f32(h, D, w, js, cw, rot=plan.contract.rot)[32:32 + cnt]
        assert_bit_exact(got, ref, "/%d window at %d" % (D, m0))
    # (2) chunking invariance across an uneven split on an output boundary
    cut = D * (total // 3)
    plan.reset()
    ya = plan.process(x[:cut])
    yb = plan.process(x[cut:])
    _sync()
    assert torch.equal(torch.view_as_real(y[: cut // D]), torch.view_as_real(ya))
    assert torch.equal(torch.view_as_real(y[cut // D:]), torch.view_as_real(yb))
    del ya, yb
    # (3) linearity under an exact scaling: a power of two commutes with every fused multiply-add
    x.mul_(0.25)
    plan.reset()
    y4 = plan.process(x)
    _sync()
    assert torch.equal(torch.view_as_real(y4) * 4.0, torch.view_as_real(y))


def test_full_size_interpolator(oracle):
    """Config 3 TX (256 taps, x8): 2^25 input samples -> 2^28 outputs."""
    import torch
    L, nt = 8, 256
    h = sxxcvr_amd.design_lowpass(nt, L, 8.0, float(L))
    n = 1 << (LOG2N - 3)
    x = torch.empty(n, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 0, 0)
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, L)
    plan.set_kernel(KERNEL_TILED)
    y = plan.process(x)
    _sync()
    assert y.numel() == n * L
    for q0 in (0, 255, 256 * 999 - 7, n // 2 + 1, n - 100):
        cnt = min(100, n - q0)
        got = to_cpu(y[q0 * L:(q0 + cnt) * L])
        if q0 < 32:
            ref = oracle.interp_f32(h, L, oracle.synth_iq(SEED, 0, 0, q0 + cnt), 2)[q0 * L:]
        else:
            ref = oracle.interp_f32(h, L, oracle.synth_iq(SEED, 0, q0 - 32, 32 + cnt), 2)[32 * L:]
        assert_bit_exact(got, ref, "x8 window at %d" % q0)
    cut = n // 3
    plan.reset()
    ya = plan.process(x[:cut])
    yb = plan.process(x[cut:])
    _sync()
    assert torch.equal(torch.view_as_real(y[: cut * L]), torch.view_as_real(ya))
    assert torch.equal(torch.view_as_real(y[cut * L:]), torch.view_as_real(yb))


def test_full_size_cf16_vs_cf32(oracle):
    """Config 5: 1024-tap /32 with fp16 IQ storage against the CF32 path on the same stream.  Stated
    tolerance: |y16 - y32| <= (2^-11 + 2^-11) * sum|h| per component (input rounded to half once: relative
    2^-11 of |x| <= 1; output rounded to half once: 2^-11 of |y| <= sum|h|), checked on the whole output."""
    import torch
    D, nt = 32, 1024
    h = sxxcvr_amd.design_lowpass(nt, D)
    n = 1 << LOG2N
    x = torch.empty(n, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 0, 0)
    p32 = sxxcvr_amd.Resampler(DECIMATE, h, D)
    y32 = torch.view_as_real(p32.process(x)).clone()
    x16 = torch.empty(n, dtype=torch.int32, device="cuda")
    sxxcvr_amd.synth_fill(x16, SEED, 0, 0, fmt="CF16")
    del x
    p16 = sxxcvr_amd.Resampler(DECIMATE, h, D, fmt="CF16")
    p16.set_kernel(KERNEL_TILED)
    y16w = p16.process(x16)
    _sync()
    y16 = y16w.view(torch.float16).float().view(-1, 2)
    bound = 2.0 ** -10 * float(np.abs(h.astype(np.float64)).sum())
    err = float((y16 - y32).abs().max().item())
    assert err <= bound, (err, bound)
    assert err > 0.0                                   # the two paths really differ in storage
    # spot window of the CF16 path against the oracle on the half-rounded input, bit for bit
    js, cw = p16.contract
    m0, cnt = (n // D) // 2 + 11, 150
    w = oracle.synth_iq(SEED, 0, D * m0 - nt, nt + D * cnt)
    wq = oracle.f16_to_f32(oracle.f32_to_f16(w.view(np.float32))).view(np.complex64)
    want = oracle.f32_to_f16(oracle.decim_f32(h, D, wq, js, cw)[32:32 + cnt].view(np.float32))
    got = to_cpu(y16w[m0:m0 + cnt]).view(np.uint16)
    assert np.array_equal(got, want)


def test_full_size_eight_channels(oracle):
    """Config 4 layout on one GPU: 8 channels x 2^25 samples in one launch == each channel on its own."""
    import torch
    h = sxxcvr_amd.design_lowpass(128, 4)
    nchan, n = 8, 1 << (LOG2N - 3)
    x = torch.empty((nchan, n), dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, first_channel=16, start=0)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, nchan=nchan)
    y = plan.process(x)
    _sync()
    assert tuple(y.shape) == (nchan, n // 4)
    single = sxxcvr_amd.Resampler(DECIMATE, h, 4)
    for c in (0, 3, 7):
        single.reset()
        yc = single.process(x[c])
        _sync()
        assert torch.equal(torch.view_as_real(yc), torch.view_as_real(y[c]))
        # and the channel is the one the source says it is: first outputs against the oracle
        ref = oracle.decim_f32(h, 4, oracle.synth_iq(SEED, 16 + c, 0, 4 * 300), 2, 4)
        assert_bit_exact(to_cpu(y[c, :300]), ref, "channel %d" % c)


@pytest.mark.parametrize("mode,ratio", [("decim", 4), ("decim", 32), ("interp", 8), ("decim", 48), ("decim", 96), ("interp", 48), ("interp", 96)])
def test_offsets_beyond_4_gib(oracle, mode, ratio):
    """One call over a buffer four times the benchmark's (2^30 wide-rate samples = 8 GiB): byte offsets no
    longer fit 32 bits anywhere in the stream.  The last outputs must still be the oracle's."""
    import torch
    nt = 32 * ratio
    wide = 1 << 30
    if mode == "decim":
        h = sxxcvr_amd.design_lowpass(nt, ratio)
        n_in = wide
        plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio)
    else:
        h = sxxcvr_amd.design_lowpass(nt, ratio, 8.0, float(ratio))
        n_in = wide // ratio
        plan = sxxcvr_amd.Resampler(INTERPOLATE, h, ratio)
    plan.set_kernel(KERNEL_TILED)
    x = torch.empty(n_in, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 0, 0)
    y = plan.process(x)
    _sync()
    if mode == "decim":
        total = n_in // ratio
        for m0 in (total // 2 + 5, total - 150):                      # input byte offsets around 4 GiB and 8 GiB
            w = oracle.synth_iq(SEED, 0, ratio * m0 - nt, nt + ratio * 150)
            ref = oracle.decim_f32(h, ratio, w, 2, 4, rot=plan.contract.rot)[32:32 + 150]
            assert_bit_exact(to_cpu(y[m0:m0 + 150]), ref, "/%d at output %d" % (ratio, m0))
    else:
        for q0 in (n_in // 2 + 3, n_in - 60):                         # output byte offsets around 4 GiB and 8 GiB
            ref = oracle.interp_f32(h, ratio, oracle.synth_iq(SEED, 0, q0 - 32, 32 + 60), 2)[32 * ratio:]
            assert_bit_exact(to_cpu(y[q0 * ratio:(q0 + 60) * ratio]), ref, "x%d at input %d" % (ratio, q0))
