"""Whole-stream parity at BASELINE.json's full sizes: EVERY output of one call over 2^28 wideband-side samples is
compared with the order-matched CPU oracle, for configs 2, 3 (RX and TX) and 5 (CF32 and CF16).

The stream is regenerated on the host (sxo_synth_iq, all cores), filtered by the oracle's multi-threaded entry
points (AVX2+FMA build of the same C source where the host has it, checked bit for bit against the portable
build on a sample first) and compared in blocks of 2^20 outputs, so that a mismatch is located.  The window
tests in test_gpu_fullsize.py stay as the fast path; these are the exhaustive ones (about a second of oracle
time each on the GPU box's 128 threads)."""
import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_TILED

pytestmark = pytest.mark.gpu

SEED = 0x51255
LOG2N = 28
BLOCK = 1 << 20


@pytest.fixture(scope="module")
def fast_oracle(oracle):
    """The oracle the whole-stream runs use: the AVX2+FMA build when the host supports it (same source, explicit
    fmaf, -ffp-contract=off: the bits of the portable build, which a sample confirms), else the portable one."""
    import oracle_lib
    cpuinfo = open("/proc/cpuinfo").read()
    if " avx2" not in cpuinfo or " fma" not in cpuinfo:
        return oracle
    fast = oracle_lib.Oracle(fast=True)
    x = oracle.synth_iq(SEED, 7, 0, 1 << 16)
    for ntaps, ratio in ((128, 4), (256, 8), (1024, 32)):
        h = oracle.design_lowpass(ntaps, ratio)
        a = oracle.decim_f32(h, ratio, x, 2, 4)
        b = fast.decim_f32(h, ratio, x, 2, 4, threads=4)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "fast oracle build differs (decimator %d/%d)" % (ntaps, ratio)
    h = oracle.design_lowpass(256, 8, 8.0, 8.0)
    a = oracle.interp_f32(h, 8, x[:8192], 2)
    b = fast.interp_f32_mt(h, 8, x[:8192], 2, threads=4)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "fast oracle build differs (interpolator)"
    return fast


def _enough_memory(gib):
    try:
        import psutil
        if psutil.virtual_memory().available < gib * (1 << 30):
            pytest.skip("needs %d GiB of host memory" % gib)
    except ImportError:
        pass


def _compare_blocks(got, ref, what):
    """got, ref: 1-D arrays of equal dtype/length holding bit patterns; reports the first differing block."""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    n = got.size
    for b0 in range(0, n, BLOCK):
        g, r = got[b0:b0 + BLOCK], ref[b0:b0 + BLOCK]
        if not np.array_equal(g, r):
            bad = np.nonzero(g != r)[0]
            raise AssertionError("%s: block %d (outputs %d..%d): %d outputs differ, first at %d: got %#x want %#x" % (
                what, b0 // BLOCK, b0, b0 + g.size - 1, bad.size, b0 + int(bad[0]), int(g[bad[0]]), int(r[bad[0]])))
    return n


@pytest.mark.parametrize("name,ntaps,ratio", [("config 2", 128, 4), ("config 3 RX", 256, 8), ("decimate by 16", 512, 16),
                                               ("config 5 CF32", 1024, 32), ("master clock / 768: decimate by 48", 1536, 48),
                                               ("master clock / 1536: decimate by 96", 3072, 96)])
def test_whole_stream_decimators_cf32(fast_oracle, name, ntaps, ratio):
    import torch
    _enough_memory(8)
    orc = fast_oracle
    n = 1 << LOG2N
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(ntaps, ratio)
    x = torch.empty(n, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 0, 0)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio)
    plan.set_kernel(KERNEL_TILED)
    y = plan.process(x)
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.uint64)
    xs = orc.synth_iq_mt(SEED, 0, 0, n, threads)
    # the stream the GPU filtered is the oracle's stream (checked on its first and last 2^20 samples and one block in between)
    for s0 in (0, (n // 3) & ~(BLOCK - 1), n - BLOCK):
        assert np.array_equal(x[s0:s0 + BLOCK].cpu().numpy().view(np.uint64), xs[s0:s0 + BLOCK].view(np.uint64)), "source"
    del x
    ref = orc.decim_f32(h, ratio, xs, 2, 4, threads=threads, rot=plan.contract.rot).view(np.uint64)
    compared = _compare_blocks(got, ref, name)
    assert compared == -(-n // ratio)
    print("%s: %d outputs compared bit for bit" % (name, compared))


def test_whole_stream_config4_slice_eight_channels(fast_oracle):
    """BASELINE config 4 as one GPU sees it: 8 independent channels x 2^25 samples in one launch (blockIdx.y =
    channel), the slice every rank of `bench.py --gpus N` runs.  Every output of every channel is compared; the
    channels are those of rank 3 (24..31) so that the channel index is not confused with the row index."""
    import torch
    _enough_memory(8)
    orc = fast_oracle
    nchan, first = 8, 24
    n = 1 << (LOG2N - 3)
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(128, 4)
    x = torch.empty((nchan, n), dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, first, 0)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, nchan=nchan)
    y = plan.process(x)
    torch.cuda.synchronize()
    del x
    compared = 0
    for c in range(nchan):
        xs = orc.synth_iq_mt(SEED, first + c, 0, n, threads)
        ref = orc.decim_f32(h, 4, xs, *plan.contract, threads=threads).view(np.uint64)
        compared += _compare_blocks(y[c].cpu().numpy().view(np.uint64), ref, "config 4 slice, channel %d" % (first + c))
    assert compared == nchan * (n // 4)
    print("config 4 slice: %d outputs compared bit for bit" % compared)


@pytest.mark.parametrize("name,ratio", [("config 3 TX", 8), ("master clock / 768: interpolate by 48", 48),
                                        ("master clock / 1536: interpolate by 96", 96)])
def test_whole_stream_interpolator_config3_tx(fast_oracle, name, ratio):
    import torch
    _enough_memory(8)
    orc = fast_oracle
    ntaps = 32 * ratio
    n_in = (1 << LOG2N) // ratio
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(ntaps, ratio, 8.0, float(ratio))
    x = torch.empty(n_in, dtype=torch.complex64, device="cuda")
    sxxcvr_amd.synth_fill(x, SEED, 1, 0)
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, ratio)
    plan.set_kernel(KERNEL_TILED)
    y = plan.process(x)
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.uint64)
    del y
    xs = orc.synth_iq_mt(SEED, 1, 0, n_in, threads)
    assert np.array_equal(x.cpu().numpy().view(np.uint64), xs.view(np.uint64)), "source"
    ref = orc.interp_f32_mt(h, ratio, xs, plan.contract[0], threads=threads).view(np.uint64)
    compared = _compare_blocks(got, ref, name)
    assert compared == n_in * ratio
    print("%s: %d outputs compared bit for bit" % (name, compared))


def test_whole_stream_decimator_cf16_config5(fast_oracle):
    """CF16 storage: the GPU's half-precision stream is copied back, widened exactly to fp32 on the host, filtered
    by the oracle and rounded to half once; every output half-pair must match."""
    import torch
    _enough_memory(10)
    orc = fast_oracle
    ratio, ntaps = 32, 1024
    n = 1 << LOG2N
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(ntaps, ratio)
    x16 = torch.empty(n, dtype=torch.int32, device="cuda")               # one word = (I, Q) as IEEE halves
    sxxcvr_amd.synth_fill(x16, SEED, 2, 0, fmt="CF16")
    plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio, fmt="CF16")
    plan.set_kernel(KERNEL_TILED)
    y = plan.process(x16)
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.uint32)
    words = x16.cpu().numpy()
    del x16
    # the source in half precision is the oracle's source rounded once (sample of 2^20), then widened exactly
    probe = orc.f32_to_f16(orc.synth_iq(SEED, 2, 0, BLOCK).view(np.float32)).view(np.uint32)
    assert np.array_equal(words[:BLOCK].view(np.uint32), probe), "CF16 source"
    xq = words.view(np.float16).astype(np.float32).view(np.complex64)
    del words
    ref32 = orc.decim_f32(h, ratio, xq, *plan.contract, threads=threads)
    ref = orc.f32_to_f16(ref32.view(np.float32)).view(np.uint32)
    compared = _compare_blocks(got, ref, "config 5 CF16")
    assert compared == n // ratio
    print("config 5 CF16: %d outputs compared bit for bit" % compared)


@pytest.mark.parametrize("name,ntaps,ratio", [("/4 on wire words", 128, 4), ("/8 on wire words (scalar-tap subset form)", 256, 8),
                                               ("/32 on wire words", 1024, 32)])
def test_whole_stream_decimators_on_wire_words(fast_oracle, name, ntaps, ratio):
    """Row f-3 at stream length: 2^26 S32_LE wire-word samples of the synthetic source through the decimators that convert
    them on load (convert_rx_buffer, SX.cpp:103-112), EVERY output against oracle conversion + oracle FIR."""
    import torch
    _enough_memory(6)
    orc = fast_oracle
    n = 1 << 26
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(ntaps, ratio)
    words = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    sxxcvr_amd.synth_fill(torch.view_as_complex(words.view(torch.float32)), SEED, 2, 0, fmt="S32")
    plan = sxxcvr_amd.Resampler(DECIMATE, h, ratio, fmt="S32")
    plan.set_kernel(KERNEL_TILED)
    y = plan.process(words)
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.uint64)
    w = words.cpu().numpy()
    del words
    xs = np.empty(n, dtype=np.complex64)
    orc.convert_rx_into(w.ravel(), xs.view(np.float32), threads)
    ref = orc.decim_f32(h, ratio, xs, *plan.contract, threads=threads).view(np.uint64)
    compared = _compare_blocks(got, ref, name)
    assert compared == n // ratio
    print("%s: %d outputs compared bit for bit" % (name, compared))


def test_whole_stream_interpolator_to_wire_words(fast_oracle):
    """x8 to S32_LE wire words with the keying bits (convert_tx_buffer, SX.cpp:116-137) through interp8_pass_kernel<2, ., S32OUT>:
    2^26 output words pairs, every one against oracle FIR + oracle conversion; the input clips in places and falls under the
    keying threshold in others."""
    import torch
    _enough_memory(6)
    orc = fast_oracle
    ratio, ntaps = 8, 256
    n_in = (1 << 26) // ratio
    threads = orc.max_threads()
    h = sxxcvr_amd.design_lowpass(ntaps, ratio, 8.0, float(ratio))
    xs = (orc.synth_iq_mt(SEED, 3, 0, n_in, threads) * np.float32(1.3)).astype(np.complex64)
    xs[n_in // 5:n_in // 4] *= np.float32(1e-4)
    thr2 = np.float32(1e-3) * np.float32(1e-3)
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, ratio, fmt="S32")
    plan.set_kernel(KERNEL_TILED)
    plan.set_tx_threshold(float(thr2))
    y = plan.process(torch.from_numpy(xs).cuda())
    torch.cuda.synchronize()
    got = y.cpu().numpy().reshape(-1, 2).view(np.uint64).ravel()
    ref = orc.convert_tx(orc.interp_f32_mt(h, ratio, xs, plan.contract[0], threads=threads), thr2).reshape(-1, 2).view(np.uint64).ravel()
    compared = _compare_blocks(got, ref, "x8 to wire words")
    assert compared == 1 << 26
    keyed = (ref & np.uint64(3)) == 3
    assert keyed.any() and (~keyed).any()
