"""Every selectable variant of the decimate-by-4 tile kernel (tools/kbench.py's A/B knobs) must
produce the bits of the default kernel: double-buffered LDS-DMA, contiguous-run schedule, packed
FMA arithmetic, SGPR-resident taps, different wave counts, and the second-generation tile kernel
(deferred stores, separate tap fetch, multi-wave workgroups).  These variants live in the PROFILING
build of the library (libsxfir_prof.so, Resampler(profiling=True)); the production library has no
knobs, which test_production_library_ignores_the_environment checks."""
import os

import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_TILED
from gpu_util import assert_bit_exact, to_cpu, to_gpu

pytestmark = pytest.mark.gpu

KNOBS = ("SXFIR_TILE_VARIANT", "SXFIR_OVERSUB", "SXFIR_OCC", "SXFIR_ABLATE", "SXFIR_SCHED", "SXFIR_LDS_PAD", "SXFIR_DENSE_NT",
         "SXFIR_DENSE", "SXFIR_MULTI_W", "SXFIR_MULTI_PS", "SXFIR_DENSE_HC")


@pytest.mark.parametrize("env", [
    {"SXFIR_TILE_VARIANT": "db"},
    {"SXFIR_TILE_VARIANT": "db", "SXFIR_SCHED": "1", "SXFIR_OVERSUB": "1"},
    {"SXFIR_SCHED": "1", "SXFIR_OVERSUB": "1"},
    {"SXFIR_OVERSUB": "3", "SXFIR_OCC": "5"},
    {"SXFIR_ABLATE": "3"},
    {"SXFIR_TILE_VARIANT": "sg"},
    {"SXFIR_TILE_VARIANT": "sg4"},
    {"SXFIR_TILE_VARIANT": "pair"},
    {"SXFIR_TILE_VARIANT": "pair", "SXFIR_OVERSUB": "3"},
    {"SXFIR_TILE_VARIANT": "t2:1:320"},
    {"SXFIR_TILE_VARIANT": "t2:1:576"},
    {"SXFIR_TILE_VARIANT": "t2:1:576", "SXFIR_OVERSUB": "64"},
    {"SXFIR_TILE_VARIANT": "wide"},
    {"SXFIR_TILE_VARIANT": "wide", "SXFIR_OVERSUB": "3"},
    {"SXFIR_TILE_VARIANT": "wident12"},
    {"SXFIR_TILE_VARIANT": "wident32", "SXFIR_OVERSUB": "5"},
    {"SXFIR_TILE_VARIANT": "widentp24"},
    {"SXFIR_TILE_VARIANT": "widepol200"},
    {"SXFIR_TILE_VARIANT": "widepol310", "SXFIR_OVERSUB": "5"},
    {"SXFIR_TILE_VARIANT": "widepol11"},
    {"SXFIR_TILE_VARIANT": "t2s"},                                   # round 3's shipped form, now the A/B partner
    {"SXFIR_TILE_VARIANT": "t2s", "SXFIR_OVERSUB": "3"},
    {"SXFIR_TILE_VARIANT": "t2:1:525376"},                           # ... and its option-bit spelling, nt loads
    {"SXFIR_TILE_VARIANT": "t2:1:66624", "SXFIR_OVERSUB": "64"},
    {"SXFIR_TILE_VARIANT": "t2:1:197696"},
    {"SXFIR_TILE_VARIANT": "t2:16:66752", "SXFIR_OVERSUB": "1"},
    {"SXFIR_TILE_VARIANT": "wide", "SXFIR_OVERSUB": "64", "SXFIR_SCHED": "2"},
    {"SXFIR_TILE_VARIANT": "pairx"},
    {"SXFIR_TILE_VARIANT": "pairx", "SXFIR_OVERSUB": "5"},
    {"SXFIR_TILE_VARIANT": "pair", "SXFIR_OVERSUB": "64", "SXFIR_SCHED": "2"},
] + [{"SXFIR_TILE_VARIANT": "t2:%d:%d" % (w, o), "SXFIR_OVERSUB": ov, "SXFIR_SCHED": sc}
     for (w, o, ov, sc) in [(1, 0, "16", "0"), (1, 1, "16", "0"), (1, 2, "64", "2"), (1, 3, "3", "0"), (1, 4, "2", "0"),
                            (1, 5, "16", "0"), (1, 6, "1", "2"), (1, 7, "16", "0"), (1, 9, "16", "0"), (1, 11, "7", "0"),
                            (2, 0, "16", "0"), (2, 1, "5", "2"), (2, 2, "64", "2"), (2, 3, "16", "0"), (2, 6, "16", "0"),
                            (2, 7, "3", "0"), (4, 2, "64", "2"), (4, 3, "16", "0"), (4, 7, "16", "0"), (8, 2, "64", "2"),
                            (8, 3, "16", "0"), (1, 16, "16", "0"), (1, 17, "16", "0"), (1, 19, "5", "0"), (2, 17, "16", "0"),
                            (2, 19, "64", "0"), (1, 32, "16", "0"), (1, 33, "16", "0"), (2, 35, "16", "0"), (1, 49, "3", "0"),
                            (2, 51, "16", "0"), (1, 64, "16", "0"), (1, 65, "16", "0"), (1, 68, "4", "0"), (1, 69, "16", "0"),
                            (1, 80, "16", "0"), (1, 81, "7", "0"), (2, 64, "64", "2"), (2, 65, "16", "0"), (4, 65, "16", "0"),
                            (16, 192, "1", "0"), (16, 193, "1", "0"), (16, 128, "1", "0"), (8, 192, "1", "0"), (4, 192, "2", "0"),
                            (16, 224, "1", "2"), (16, 192, "3", "0")]])
def test_variant_matches_oracle(oracle, monkeypatch, env):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = sxxcvr_amd.design_lowpass(128, 4)
    n = (1 << 20) + 4 * 77
    x = oracle.synth_iq(0x51255, 4, 0, n + 4096)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, profiling=True)     # knobs are read at plan creation
    plan.set_kernel(KERNEL_TILED)
    y1 = to_cpu(plan.process(to_gpu(x[:n])))
    y2 = to_cpu(plan.process(to_gpu(x[n:])))                # exercises the fused history carry-over
    ref = oracle.decim_f32(h, 4, x, 2, 4)
    assert_bit_exact(np.concatenate([y1, y2]), ref, "variant %r" % env)


@pytest.mark.gpu
def test_short_tail_schedule_matches_oracle(oracle, monkeypatch):
    """SXFIR_SCHED=3: the launch ends with one-tile waves (the last CU-filling set of long waves is replaced by as
    many short ones as it had tiles).  2^24 samples = 16384 tiles with two generations of waves, so that the split
    applies; two calls, so that the short wave that owns the last tile carries the history over."""
    import torch
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_TILE_VARIANT", "t2s")          # the schedule belongs to the 4-outputs-per-lane kernel
    monkeypatch.setenv("SXFIR_SCHED", "3")
    monkeypatch.setenv("SXFIR_OVERSUB", "2")
    h = sxxcvr_amd.design_lowpass(128, 4)
    n = 1 << 24
    x = oracle.synth_iq_mt(0x51255, 9, 0, 2 * n, 8)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    y1 = to_cpu(plan.process(to_gpu(x[:n])))
    y2 = to_cpu(plan.process(to_gpu(x[n:])))
    ref = oracle.decim_f32(h, 4, x, 2, 4, threads=8)
    assert_bit_exact(np.concatenate([y1, y2]), ref, "short-tail schedule")


@pytest.mark.gpu
@pytest.mark.parametrize("D,n_in", [(4, 70000), (8, 1 << 16), (16, 50000), (32, 4096 * 5 + 32 * 3)])
def test_quarter_row_split_variant(oracle, monkeypatch, D, n_in):
    """SXFIR_MULTI_PS=4: the multi-column decimator with the 32 tap rows split over four lanes.  The plan
    then reports the contract (4, 4) and must match the oracle run with exactly that contract, across a
    call boundary (fused history carry-over), CF32 and CF16."""
    import sxxcvr_amd
    from sxxcvr_amd.resampler import DECIMATE, KERNEL_TILED
    monkeypatch.setenv("SXFIR_MULTI_PS", "4")
    if D == 4:
        monkeypatch.setenv("SXFIR_TILE_VARIANT", "mu")      # route /4 through the multi-column kernel
    n_in -= n_in % D
    h = sxxcvr_amd.design_lowpass(32 * D, D)
    x = oracle.synth_iq(0x51255, 21, 0, n_in + 2048)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, D, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    assert plan.contract == (4, 4)
    y = np.concatenate([to_cpu(plan.process(to_gpu(x[:n_in]))), to_cpu(plan.process(to_gpu(x[n_in:])))])
    assert_bit_exact(y, oracle.decim_f32(h, D, x, 4, 4), "quarter split D=%d" % D)
    if D != 4:
        x16 = oracle.f32_to_f16(x.view(np.float32))
        xq = oracle.f16_to_f32(x16).view(np.complex64)
        p16 = sxxcvr_amd.Resampler(DECIMATE, h, D, fmt="CF16", profiling=True)
        p16.set_kernel(KERNEL_TILED)
        assert p16.contract == (4, 4)
        got = to_cpu(p16.process(to_gpu(x16.view(np.uint32).view(np.int32)))).view(np.uint16)
        want = oracle.f32_to_f16(oracle.decim_f32(h, D, xq, 4, 4).view(np.float32))
        assert np.array_equal(got, want), "quarter split CF16 D=%d" % D


def test_production_library_ignores_the_environment(oracle, monkeypatch):
    """A drop-in driver must not change its arithmetic because an environment variable is set: with the
    profiling knobs in the environment (one of them an ablation mode that gives wrong results in the
    profiling build) the production library still produces the oracle's bits with its default contract."""
    monkeypatch.setenv("SXFIR_ABLATE", "1")
    monkeypatch.setenv("SXFIR_TILE_VARIANT", "sg")
    monkeypatch.setenv("SXFIR_OVERSUB", "3")
    monkeypatch.setenv("SXFIR_MULTI_PS", "4")
    for D in (4, 8):
        h = sxxcvr_amd.design_lowpass(32 * D, D)
        n = 1 << 18
        x = oracle.synth_iq(0x51255, 9, 0, n)
        plan = sxxcvr_amd.Resampler(DECIMATE, h, D)
        plan.set_kernel(KERNEL_TILED)
        assert plan.contract == (2, 4)
        assert_bit_exact(to_cpu(plan.process(to_gpu(x))), oracle.decim_f32(h, D, x, 2, 4), "production D=%d" % D)


@pytest.mark.parametrize("opt", [192, 128])
def test_cu_queue_variant_multichannel(oracle, monkeypatch, opt):
    """One 16-wave workgroup per CU taking tiles from its LDS queue: three channels, several calls, a ragged last
    call, history carried per channel by whichever wave gets the channel's last tile."""
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_TILE_VARIANT", "t2:16:%d" % opt)
    monkeypatch.setenv("SXFIR_OVERSUB", "1")
    h = sxxcvr_amd.design_lowpass(128, 4)
    nchan, lens = 3, [1 << 18, 8 * 1000, (1 << 16) + 8 * 33, 8 * 5]      # even output counts: tiled kernel in every call
    total = sum(lens)
    xs = np.stack([oracle.synth_iq(0x51255, 30 + c, 0, total) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(xs[:, pos:pos + n]))))
        pos += n
    got = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(got[c], oracle.decim_f32(h, 4, xs[c], 2, 4), "CU queue variant %d channel %d" % (opt, c))


@pytest.mark.gpu
@pytest.mark.parametrize("ipass,oversub", [("0", "4"), ("1", "1"), ("1", "16"), ("4", "4"), ("4", "1"), ("1w0", "1"), ("1w0", "16")])
def test_x8_interpolator_forms_match_oracle(oracle, monkeypatch, ipass, oversub):
    """x8, 256 taps: interp8_pass_kernel (scalar taps; two inputs per lane as shipped, or four) and interp_tile_kernel (taps in
    VGPRs, SXFIR_IPASS=0) give the oracle's bits: streaming over three calls -- a first tile that takes its history from
    the plan, interior tiles with the prefetch and its counted wait, a ragged last tile -- and several channels."""
    import torch
    for k in KNOBS + ("SXFIR_IPASS", "SXFIR_IPASS_WAIT0"):
        monkeypatch.delenv(k, raising=False)
    if ipass.endswith("w0"):                # the vmcnt(0) build of the shipped form: the counted wait's A/B partner
        ipass = ipass[:-2]
        monkeypatch.setenv("SXFIR_IPASS_WAIT0", "1")
    monkeypatch.setenv("SXFIR_IPASS", ipass)
    monkeypatch.setenv("SXFIR_OVERSUB", oversub)
    h = sxxcvr_amd.design_lowpass(256, 8, 8.0, 8.0)
    nchan, lens = 2, [256 * 300 + 77, 5, 256 * 64]
    x = np.stack([oracle.synth_iq(0x51255, 30 + c, 0, sum(lens)) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, 8, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(np.ascontiguousarray(x[:, pos:pos + n])))))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(y[c], oracle.interp_f32(h, 8, x[c], 2), "x8 form %s channel %d" % (ipass, c))


@pytest.mark.gpu
@pytest.mark.parametrize("ipass,oversub", [("0", "4"), ("0", "16"), ("1", "1"), ("1", "8"), ("1", "16")])
def test_x4_interpolator_forms_match_oracle(oracle, monkeypatch, ipass, oversub):
    """x4, 128 taps: interp8_pass_kernel<4 inputs per lane, L = 4> (scalar taps, two passes: shipped since round 5) and
    interp_tile_kernel<4> (taps in VGPRs, SXFIR_IPASS=0: its A/B partner) give the oracle's bits: streaming over three calls --
    a first tile that takes its history from the plan, interior tiles with the prefetch and its counted wait, a ragged
    last tile -- and several channels."""
    for k in KNOBS + ("SXFIR_IPASS", "SXFIR_IPASS_WAIT0"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_IPASS", ipass)
    monkeypatch.setenv("SXFIR_OVERSUB", oversub)
    h = sxxcvr_amd.design_lowpass(128, 4, 8.0, 4.0)
    nchan, lens = 2, [256 * 300 + 77, 5, 256 * 64 + 130]
    x = np.stack([oracle.synth_iq(0x51255, 33 + c, 0, sum(lens)) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, 4, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(np.ascontiguousarray(x[:, pos:pos + n])))))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(y[c], oracle.interp_f32(h, 4, x[c], 2), "x4 form %s channel %d" % (ipass, c))


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [16, 32, 48, 96])
@pytest.mark.parametrize("ipass,oversub", [("0", "2"), ("1", "1"), ("1", "8"), ("1", "16")])
def test_x16_to_x96_interpolator_forms_match_oracle(oracle, monkeypatch, ratio, ipass, oversub):
    """x16 .. x96: interp8_pass_kernel over phase blocks of sixteen (scalar taps, one window per tile for all blocks, the partial last
    round of tiles dealt plainly: shipped since round 5) and interp_tile_kernel (taps in VGPRs; x48 / x96 as three phase blocks;
    SXFIR_IPASS=0: the A/B partner) give the oracle's bits: streaming over calls with first, interior and ragged last tiles at
    several generation counts (whole and partial rounds of tiles), two channels."""
    for k in KNOBS + ("SXFIR_IPASS", "SXFIR_IPASS_WAIT0"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_IPASS", ipass)
    monkeypatch.setenv("SXFIR_OVERSUB", oversub)
    h = sxxcvr_amd.design_lowpass(32 * ratio, ratio, 8.0, float(ratio))
    nchan, lens = 2, [128 * 2500 + 77, 5, 128 * 64 + 130, 1]
    x = np.stack([oracle.synth_iq(0x51255, 36 + c, 0, sum(lens)) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, ratio, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(np.ascontiguousarray(x[:, pos:pos + n])))))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(y[c], oracle.interp_f32(h, ratio, x[c], 2), "x%d form %s channel %d" % (ratio, ipass, c))


@pytest.mark.gpu
@pytest.mark.parametrize("subset", ["1", "0"])
@pytest.mark.parametrize("nchan,lens", [(1, [512 * 40 + 8 * 3, 8 * 5, 512 * 9]), (3, [512 * 7 + 8 * 77, 1 << 17])])
def test_div8_forms_match_oracle(oracle, monkeypatch, nchan, lens, subset):
    """/8, 256 taps, CF32: decim_dense_kernel<8, ..., SUBSET> (the shipped form: the four tap subsets on the four waves,
    taps in SGPRs, partial sums exchanged through LDS) and the VGPR-tap form (SXFIR_DENSE_SUBSET=0) give the oracle's
    bits: streaming over several calls with ragged tails and several channels."""
    for k in KNOBS + ("SXFIR_DENSE_SUBSET",):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_DENSE_SUBSET", subset)
    h = sxxcvr_amd.design_lowpass(256, 8)
    total = sum(lens)
    x = np.stack([oracle.synth_iq(0x51255, 40 + c, 0, 8 * total) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 8, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(np.ascontiguousarray(x[:, 8 * pos:8 * (pos + n)])))))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(y[c], oracle.decim_f32(h, 8, x[c], 2, 4), "/8 subset form, channel %d" % c)


@pytest.mark.gpu
@pytest.mark.parametrize("subset", ["1", "0"])
@pytest.mark.parametrize("nchan,lens", [(1, [512 * 12 + 8 * 3, 8 * 5, 512 * 9]), (2, [512 * 7 + 8 * 77, 1 << 16])])
def test_div8_forms_on_wire_words_match_oracle(oracle, monkeypatch, nchan, lens, subset):
    """/8, 256 taps, S32_LE wire words (convert_rx_buffer, SX.cpp:103-112, inside the kernel): the scalar-tap form (taps
    times 2^-31 in SGPRs, shipped) and the VGPR-tap form give the bits of oracle conversion + oracle FIR; several calls
    with ragged tails, several channels, extreme words at the start."""
    import torch
    for k in KNOBS + ("SXFIR_DENSE_SUBSET",):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_DENSE_SUBSET", subset)
    h = sxxcvr_amd.design_lowpass(256, 8)
    total = sum(lens)
    rng = np.random.default_rng(77 + nchan)
    words = rng.integers(-2 ** 31, 2 ** 31, size=(nchan, 8 * total, 2), dtype=np.int64).astype(np.int32)
    words[:, :3, :] = np.array([[2 ** 31 - 1, -2 ** 31], [1, -1], [0, 0x7FFFFF80]], dtype=np.int32)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 8, nchan=nchan, fmt="S32", profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        blk = torch.from_numpy(np.ascontiguousarray(words[:, 8 * pos:8 * (pos + n)])).cuda()
        outs.append(to_cpu(plan.process(blk if nchan > 1 else blk[0])).reshape(nchan, -1))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        ref = oracle.decim_f32(h, 8, oracle.convert_rx(words[c].ravel()), 2, 4)
        assert_bit_exact(y[c], ref, "/8 on wire words, subset %s, channel %d" % (subset, c))


@pytest.mark.gpu
@pytest.mark.parametrize("ipass", ["1", "0", "1w0"])
def test_x8_interpolator_forms_to_wire_words_match_oracle(oracle, monkeypatch, ipass):
    """x8 to S32_LE wire words with the keying bits (convert_tx_buffer, SX.cpp:116-137): interp8_pass_kernel<2, KEYED, S32OUT>
    (shipped) and interp_tile_kernel<8, S32OUT> (SXFIR_IPASS=0) give the oracle's words -- clipping samples, samples under the
    keying threshold, three calls with a ragged tail, two channels -- and the keying count of the input rides along."""
    import torch
    for k in KNOBS + ("SXFIR_IPASS", "SXFIR_IPASS_WAIT0"):
        monkeypatch.delenv(k, raising=False)
    if ipass.endswith("w0"):                # <2, KEYED, S32OUT, COUNTED = false>
        ipass = ipass[:-2]
        monkeypatch.setenv("SXFIR_IPASS_WAIT0", "1")
    monkeypatch.setenv("SXFIR_IPASS", ipass)
    h = sxxcvr_amd.design_lowpass(256, 8, 8.0, 8.0)
    nchan, lens = 2, [128 * 41 + 77, 5, 128 * 64]
    x = np.stack([(oracle.synth_iq(0x51255, 50 + c, 0, sum(lens)) * np.float32(1.3)).astype(np.complex64) for c in range(nchan)])
    x[:, 300:900] *= np.float32(1e-4)
    thr2 = np.float32(1e-3) * np.float32(1e-3)
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, 8, nchan=nchan, fmt="S32", profiling=True)
    plan.set_kernel(KERNEL_TILED)
    plan.set_tx_threshold(float(thr2))
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    outs, pos, want_count = [], 0, 0
    keyed_in = (oracle.convert_tx(x[0], thr2).reshape(-1, 2)[:, 0] & 3) == 3
    for n in lens:
        blk = to_gpu(np.ascontiguousarray(x[:, pos:pos + n]))
        out = torch.empty((nchan, 8 * n, 2), dtype=torch.int32, device="cuda")
        got = plan.interpolate_keyed_ptr(blk.data_ptr(), n, n, out.data_ptr(), 8 * n, 0, n, counter.data_ptr(), st)
        assert got == 8 * n
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().reshape(nchan, -1))
        want_count += int(keyed_in[pos:pos + n].sum())
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        want = oracle.convert_tx(oracle.interp_f32(h, 8, x[c], 2), thr2)
        assert np.array_equal(y[c], want), "x8 to wire words, form %s, channel %d" % (ipass, c)
        keyed = (want[0::2] & 3) == 3
        assert keyed.any() and (~keyed).any()
    assert int(counter.item()) == want_count


@pytest.mark.gpu
@pytest.mark.parametrize("D,oversub", [(32, "8"), (32, "1"), (32, "64"), (16, "8"), (16, "3")])
def test_dense_halo_carry_matches_oracle(oracle, monkeypatch, D, oversub):
    """decim_dense_kernel<D, ..., HC> (SXFIR_DENSE_HC=1): runs of consecutive tiles per workgroup, the next tile's halo copied inside
    LDS instead of fetched again.  Streaming over several calls -- a first call whose tile 0 takes its halo from the plan's history,
    long interior runs (carried tiles), a ragged last tile, a call shorter than one tile -- and several channels, against the oracle."""
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_DENSE_HC", "1")
    monkeypatch.setenv("SXFIR_OVERSUB", oversub)
    T = 128 if D == 32 else 256
    h = sxxcvr_amd.design_lowpass(32 * D, D)
    nchan, lens = 2, [T * 700 + 6, 4, T * 64, T * 1500 + T // 2]      # even output counts: 16-byte aligned channel rows
    x = np.stack([oracle.synth_iq(0x51255, 60 + c, 0, D * sum(lens)) for c in range(nchan)])
    plan = sxxcvr_amd.Resampler(DECIMATE, h, D, nchan=nchan, profiling=True)
    plan.set_kernel(KERNEL_TILED)
    outs, pos = [], 0
    for n in lens:
        outs.append(to_cpu(plan.process(to_gpu(np.ascontiguousarray(x[:, D * pos:D * (pos + n)])))))
        pos += n
    y = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert_bit_exact(y[c], oracle.decim_f32(h, D, x[c], 2, 4), "/%d halo carry, channel %d" % (D, c))


@pytest.mark.gpu
@pytest.mark.parametrize("D,fmt,nchan", [(48, "CF32", 1), (96, "CF32", 1), (96, "CF32", 3), (48, "CF16", 2), (96, "S32", 1)])
def test_blocks_split_and_walking_forms_agree(oracle, monkeypatch, D, fmt, nchan):
    """decim_blocks_kernel (/48, /96): the form that deals (tile, block) items to workgroups and joins the block values through
    HBM (SPLIT, round 6: what every call of at most eight times as many tiles as the chip has workgroup slots runs) against the form
    in which one workgroup walks a tile's blocks (SXFIR_BLOCKS_SPLIT=0 in the profiling build).  The join adds the block values
    in the contract's order whatever the arrival order, so the two must agree bit for bit: a streamed sequence of calls -- one
    output, one tile, many tiles with a ragged tail, several hundred tiles (several rounds of items), a short one again --
    per channel; the first calls also against the oracle."""
    import torch
    for k in KNOBS + ("SXFIR_BLOCKS_SPLIT",):
        monkeypatch.delenv(k, raising=False)
    h = (np.random.default_rng(D + 1).standard_normal(32 * D) / 64.0).astype(np.float32)      # asymmetric taps
    lens = [4, 512, 512 * 7 + 76, 512 * 700 + 12, 68]          # outputs per call (multiples of four: 16-byte aligned channel rows, CF16 too)
    total = D * sum(lens)
    if fmt == "S32":
        rng = np.random.default_rng(5)
        words = rng.integers(-2 ** 31, 2 ** 31, size=(nchan, total, 2), dtype=np.int64).astype(np.int32)
        x_dev = to_gpu(words)
    elif fmt == "CF16":
        x32 = np.stack([oracle.synth_iq(0x51255, 90 + c, 0, total) for c in range(nchan)])
        halves = oracle.f32_to_f16(x32.view(np.float32))
        x_dev = to_gpu(halves.view(np.int32).reshape(nchan, total))
    else:
        x32 = np.stack([oracle.synth_iq(0x51255, 90 + c, 0, total) for c in range(nchan)])
        x_dev = to_gpu(x32)
    results = {}
    for split in ("1", "0"):
        monkeypatch.setenv("SXFIR_BLOCKS_SPLIT", split)
        plan = sxxcvr_amd.Resampler(DECIMATE, h, D, nchan=nchan, fmt=fmt, profiling=True)
        plan.set_kernel(KERNEL_TILED)
        g = plan.geometry(D * 512 * 7)
        assert g["kernel"] == "decim_blocks_kernel" and g["split"] == (D // 16 if split == "1" else 1), g
        outs, pos = [], 0
        for n in lens:
            blk = x_dev[:, D * pos:D * (pos + n)].contiguous()
            y = plan.process(blk if nchan > 1 else blk[0])
            torch.cuda.synchronize()
            outs.append(to_cpu(y).reshape(nchan, -1) if fmt != "S32" else to_cpu(y).reshape(nchan, -1))
            pos += n
        results[split] = np.concatenate(outs, axis=1)
        plan.close()
    a, b = results["1"], results["0"]
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
        "/%d %s: %d words differ between the dealt and the walking form" % (D, fmt, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
    if fmt == "CF32":
        n_chk = sum(lens[:3])
        for c in range(nchan):
            ref = oracle.decim_f32(h, D, x32[c][:D * n_chk], 2, 4, rot=1)
            assert_bit_exact(a[c][:n_chk], ref, "/%d dealt form, channel %d" % (D, c))


@pytest.mark.gpu
@pytest.mark.parametrize("L,fmt,nchan", [(32, "CF32", 1), (48, "CF32", 2), (96, "CF32", 1), (96, "S32", 1)])
def test_pass_kernel_split_and_walking_forms_agree(oracle, monkeypatch, L, fmt, nchan):
    """interp8_pass_kernel over phase blocks (x32, x48, x96): the form that deals (tile, phase block) items (PBSPLIT, round 6: what
    every call of at most four times as many tiles as the chip holds waves runs) against the form in which one wave walks a tile's
    blocks (SXFIR_IPASS_SPLIT=0 in the profiling build): a streamed sequence of calls -- one input, one tile, many tiles with a
    ragged tail, a few thousand tiles -- with and without the keying count; bit for bit against each other, the first calls
    against the oracle."""
    import torch
    for k in KNOBS + ("SXFIR_IPASS", "SXFIR_IPASS_WAIT0", "SXFIR_IPASS_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    h = (np.random.default_rng(L + 2).standard_normal(32 * L) / 8.0).astype(np.float32)
    lens = [1, 128, 128 * 9 + 77, 128 * 3000 + 5, 40]           # inputs per call
    total = sum(lens)
    x = np.stack([oracle.synth_iq(0x51255, 40 + c, 0, total) * np.float32(0.9) for c in range(nchan)])
    x_dev = to_gpu(x)
    thr2 = np.float32(0.25)
    results, counts = {}, {}
    for split in ("1", "0"):
        monkeypatch.setenv("SXFIR_IPASS_SPLIT", split)
        plan = sxxcvr_amd.Resampler(INTERPOLATE, h, L, nchan=nchan, fmt=fmt, profiling=True)
        plan.set_kernel(KERNEL_TILED)
        plan.set_tx_threshold(float(thr2))
        g = plan.geometry(128 * 9)
        assert g["kernel"] == "interp8_pass_kernel" and g["split"] == (L // 16 if split == "1" else 1), g
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        outs, pos = [], 0
        for i, n in enumerate(lens):
            blk = x_dev[:, pos:pos + n].contiguous()
            y = torch.empty((nchan, n * L), dtype=torch.complex64, device="cuda")
            if i % 2 == 0:
                plan.interpolate_keyed_ptr(blk.data_ptr(), n, n, y.data_ptr(), n * L, 0, n, counter.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
            else:
                plan.process_ptr(blk.data_ptr(), n, n, y.data_ptr(), n * L, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs.append(to_cpu(y))
            pos += n
        results[split] = np.concatenate(outs, axis=1)
        counts[split] = int(counter.item())
        plan.close()
    a, b = results["1"], results["0"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "x%d %s: the dealt and the walking form differ" % (L, fmt)
    keyed_inputs = np.concatenate([x[0][sum(lens[:i]):sum(lens[:i + 1])] for i in range(len(lens)) if i % 2 == 0])
    want = int(np.count_nonzero(keyed_inputs.real.astype(np.float32) ** 2 + keyed_inputs.imag.astype(np.float32) ** 2 >= thr2))
    assert counts["1"] == counts["0"] == want, (counts, want)
    n_chk = sum(lens[:3])
    for c in range(nchan):
        ref = oracle.interp_f32(h, L, x[c][:n_chk], 2)
        if fmt == "S32":
            assert np.array_equal(a[c][:n_chk * L].view(np.int32), oracle.convert_tx(ref, thr2)), "x%d to wire words, channel %d" % (L, c)
        else:
            assert_bit_exact(a[c][:n_chk * L], ref, "x%d dealt form, channel %d" % (L, c))


@pytest.mark.gpu
@pytest.mark.parametrize("D,split", [(48, "1"), (96, "1"), (96, "0")])
def test_blocks_kernel_waves_by_column_group_agree(oracle, monkeypatch, D, split):
    """decim_blocks_kernel<..., RP> (what ships since round 6: the block's subsets dealt to the waves by column group, the rows the
    two windows share kept in registers, P0 + P1 added in the lane) against round 5's form (SXFIR_BLOCKS_RP=0, profiling build: waves
    by row half) and the oracle: same contract, same bits; the dealt (SPLIT) and the walking instance."""
    import torch
    for k in KNOBS + ("SXFIR_BLOCKS_SPLIT", "SXFIR_BLOCKS_RP"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SXFIR_BLOCKS_SPLIT", split)
    h = (np.random.default_rng(D + 7).standard_normal(32 * D) / 64.0).astype(np.float32)
    lens = [2, 512, 512 * 9 + 76, 512 * 300 + 12]
    x = oracle.synth_iq(0x51255, 33, 0, D * sum(lens))
    x_dev = to_gpu(x)
    res = {}
    for rp in ("0", "1"):
        monkeypatch.setenv("SXFIR_BLOCKS_RP", rp)
        plan = sxxcvr_amd.Resampler(DECIMATE, h, D, profiling=True)
        plan.set_kernel(KERNEL_TILED)
        outs, pos = [], 0
        for n in lens:
            outs.append(to_cpu(plan.process(x_dev[D * pos:D * (pos + n)])))
            pos += n
        torch.cuda.synchronize()
        res[rp] = np.concatenate(outs)
        plan.close()
    assert np.array_equal(res["0"].view(np.uint32), res["1"].view(np.uint32)), "/%d: waves by column group differ from the shipped form" % D
    n_chk = sum(lens[:3])
    assert_bit_exact(res["1"][:n_chk], oracle.decim_f32(h, D, x[:D * n_chk], 2, 4, rot=1), "/%d waves by column group" % D)
