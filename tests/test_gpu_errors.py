"""Error behaviour of the inner C ABI (include/sxfir.h) on the GPU box: every misuse comes back as a negative code with a
message in sxfir_last_error(), never as a crash, an exception across the ABI or a silent fallback, and a plan that has seen
errors still computes the oracle's bits afterwards.  (The outer surface's error behaviour -- the reference's exceptions and
SOAPY_SDR_* codes -- is tests/test_gpu_device.py's.)"""
import ctypes as C

import numpy as np
import pytest

import sxxcvr_amd
from gpu_util import assert_bit_exact, to_cpu, to_gpu
from sxxcvr_amd.resampler import DECIMATE, INTERPOLATE, KERNEL_TILED

pytestmark = pytest.mark.gpu

EINVAL, EHIP, ENOMEM, EUNSUPPORTED, ENODEVICE = -1, -2, -3, -4, -5


@pytest.fixture(scope="module")
def lib():
    return sxxcvr_amd.load_sxfir()


def failed(lib, rc, want):
    msg = lib.sxfir_last_error()
    assert rc == want, (rc, want, msg)
    assert msg and len(msg) > 3, "an error code without a message"
    return msg


def test_create_refuses_bad_arguments(lib):
    taps = (C.c_float * 128)(*([0.01] * 128))
    plan = C.c_void_p()
    good = dict(mode=0, taps=taps, ntaps=128, ratio=4, nchan=1, fmt=0, device=-1)

    def create(out=True, **kw):
        a = dict(good, **kw)
        return lib.sxfir_create(C.byref(plan) if out else None, a["mode"], a["taps"], a["ntaps"], a["ratio"], a["nchan"], a["fmt"],
                                a["device"])

    failed(lib, create(out=False), EINVAL)
    failed(lib, create(taps=None), EINVAL)
    assert b"mode" in failed(lib, create(mode=2), EINVAL)
    for n in (0, -5, 65537):
        assert b"ntaps" in failed(lib, create(ntaps=n), EINVAL)
    for r in (0, -1, 4097):
        assert b"ratio" in failed(lib, create(ratio=r), EINVAL)
    for ch in (0, 65536):
        assert b"nchan" in failed(lib, create(nchan=ch), EINVAL)
    assert b"format" in failed(lib, create(fmt=7), EINVAL)
    assert b"device" in failed(lib, create(device=99), EINVAL)
    # an interpolator's taps split into whole phases (SXFIR_INTERPOLATE: ntaps % ratio == 0)
    failed(lib, create(mode=1, ntaps=100, ratio=8), EINVAL)
    # nothing was created by any of these
    assert not plan.value
    assert create() == 0 and plan.value
    assert lib.sxfir_destroy(plan) == 0
    assert lib.sxfir_destroy(None) == 0                       # destroying nothing is not an error


def test_stream_calls_refuse_bad_arguments_and_leave_the_plan_intact(lib, oracle):
    import torch
    h = sxxcvr_amd.design_lowpass(128, 4)
    plan = sxxcvr_amd.Resampler(DECIMATE, h, 4, nchan=2)
    p = plan._plan
    n = 4096
    x = oracle.synth_iq(0x51255, 3, 0, 2 * n).reshape(2, n)
    xg = to_gpu(x)
    y = torch.empty((2, n // 4), dtype=torch.complex64, device="cuda")
    n_out = C.c_size_t(77)
    vp = C.c_void_p

    def dec(pl=p, src=xg.data_ptr(), cnt=n, istr=n, dst=y.data_ptr(), ostr=n // 4):
        return lib.sxfir_decimate(pl, vp(src), cnt, istr, vp(dst), ostr, C.byref(n_out), None)

    failed(lib, dec(pl=None), EINVAL)
    failed(lib, dec(src=None), EINVAL)
    failed(lib, dec(dst=None), EINVAL)
    assert b"stride" in failed(lib, dec(istr=n - 1), EINVAL)
    assert b"stride" in failed(lib, dec(ostr=n // 4 - 1), EINVAL)
    assert b"aligned" in failed(lib, dec(src=xg.data_ptr() + 4), EINVAL)          # not on a complex sample
    assert b"aligned" in failed(lib, dec(dst=y.data_ptr() + 2), EINVAL)
    assert n_out.value == 0                                    # a failed call reports no outputs
    # the wrong direction
    assert b"direction" in failed(lib, lib.sxfir_interpolate(p, vp(xg.data_ptr()), n, n, vp(y.data_ptr()), n // 4, C.byref(n_out), None),
                                  EINVAL)
    # the tiled kernel asked for explicitly, on a call it cannot take (odd output stride between channels): refused, no fallback
    plan.set_kernel(KERNEL_TILED)
    y_odd = torch.empty((2, n // 4 + 1), dtype=torch.complex64, device="cuda")
    failed(lib, dec(dst=y_odd.data_ptr(), ostr=n // 4 + 1), EUNSUPPORTED)
    plan.set_kernel(0)
    # positions, kernels, history
    failed(lib, lib.sxfir_set_position(p, -1), EINVAL)
    failed(lib, lib.sxfir_set_kernel(p, 9), EINVAL)
    failed(lib, lib.sxfir_set_kernel(None, 0), EINVAL)
    failed(lib, lib.sxfir_set_history(p, vp(xg.data_ptr()), 10, 10, None), EINVAL)          # fewer samples than the history holds
    failed(lib, lib.sxfir_set_history(p, None, 128, 128, None), EINVAL)
    failed(lib, lib.sxfir_outputs_for(p, 100, None), EINVAL)
    failed(lib, lib.sxfir_contract(None, None, None), EINVAL)
    failed(lib, lib.sxfir_position(None, None, None), EINVAL)
    failed(lib, lib.sxfir_reset(None, None), EINVAL)
    # after all that the plan still streams the oracle's bits from position 0
    consumed, produced = C.c_int64(-1), C.c_int64(-1)
    assert lib.sxfir_position(p, C.byref(consumed), C.byref(produced)) == 0 and (consumed.value, produced.value) == (0, 0)
    got = to_cpu(plan.process(xg))
    for c in range(2):
        assert_bit_exact(got[c], oracle.decim_f32(h, 4, x[c], 2, 4), "after the refused calls, channel %d" % c)


def test_keyed_interpolation_and_shape_limits(lib):
    import torch
    h = sxxcvr_amd.design_lowpass(256, 8, 8.0, 8.0)
    plan = sxxcvr_amd.Resampler(INTERPOLATE, h, 8)
    p = plan._plan
    n = 1024
    x = torch.zeros(n, dtype=torch.complex64, device="cuda")
    y = torch.empty(8 * n, dtype=torch.complex64, device="cuda")
    counter = torch.zeros(2, dtype=torch.int64, device="cuda")
    n_out = C.c_size_t(0)
    vp = C.c_void_p

    def keyed(first=0, count=n, ctr=counter.data_ptr(), pl=p):
        return lib.sxfir_interpolate_keyed(pl, vp(x.data_ptr()), n, n, vp(y.data_ptr()), 8 * n, C.byref(n_out), first, count, vp(ctr), None)

    assert b"range" in failed(lib, keyed(first=n + 1, count=0), EINVAL)
    assert b"range" in failed(lib, keyed(first=10, count=n), EINVAL)
    failed(lib, keyed(ctr=None), EINVAL)
    assert b"aligned" in failed(lib, keyed(ctr=counter.data_ptr() + 4), EINVAL)
    dplan = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(128, 4), 4)
    failed(lib, keyed(pl=dplan._plan), EINVAL)                 # not an interpolator
    hplan = sxxcvr_amd.Resampler(INTERPOLATE, h, 8, fmt="CF16")
    failed(lib, keyed(pl=hplan._plan), EUNSUPPORTED)           # the keying rule is defined on CF32 input
    assert keyed() == 0 and n_out.value == 8 * n               # and the good call goes through
    # a shape no tiled kernel takes: asking for the tiled kernel is refused at once, the generic one serves it
    odd = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(100, 5), 5)
    failed(lib, lib.sxfir_set_kernel(odd._plan, KERNEL_TILED), EUNSUPPORTED)


def test_helpers_refuse_null_and_nonsense(lib):
    vp = C.c_void_p
    taps = (C.c_float * 8)()
    failed(lib, lib.sxfir_design_lowpass(0, 4, 8.0, 1.0, taps), EINVAL)
    failed(lib, lib.sxfir_design_lowpass(8, 0, 8.0, 1.0, taps), EINVAL)
    failed(lib, lib.sxfir_design_lowpass(8, 4, 8.0, 1.0, None), EINVAL)
    failed(lib, lib.sxfir_device_count(None), EINVAL)
    failed(lib, lib.sxfir_synth_fill(None, 16, 16, 1, 1, 0, 0, 0, None), EINVAL)
    failed(lib, lib.sxfir_malloc(None, 16), EINVAL)
    failed(lib, lib.sxfir_host_alloc(None, 16), EINVAL)
    failed(lib, lib.sxfir_host_register(None, 16), EINVAL)
    failed(lib, lib.sxfir_stream_create(None), EINVAL)
    failed(lib, lib.sxfir_event_create(None), EINVAL)
    failed(lib, lib.sxfir_event_record(None, None), EINVAL)
    failed(lib, lib.sxfir_event_sync(None), EINVAL)
    failed(lib, lib.sxfir_event_elapsed_ms(None, None, None), EINVAL)
    failed(lib, lib.sxfir_stream_wait_event(None, None), EINVAL)
    failed(lib, lib.sxfir_clock_probe_start(None, -1, 1000), EINVAL)
    failed(lib, lib.sxfir_clock_probe_read(None, None), EINVAL)
    dev = vp()
    assert lib.sxfir_malloc(C.byref(dev), 1 << 16) == 0
    for fn, args in ((lib.sxfir_convert_rx_s32, (None, dev, 4, None)), (lib.sxfir_convert_rx_s32, (dev, None, 4, None)),
                     (lib.sxfir_convert_tx_s32, (None, dev, 4, C.c_float(0.0), None)),
                     (lib.sxfir_count_keyed, (dev, 4, C.c_float(0.0), None, None)),
                     (lib.sxfir_cf32_to_cf16, (None, dev, 4, None)), (lib.sxfir_cf16_to_cf32, (dev, None, 4, None))):
        failed(lib, fn(*args), EINVAL)
    failed(lib, lib.sxfir_synth_fill(dev, 16, 16, 1, 1, 0, 0, 9, None), EINVAL)            # no such format
    failed(lib, lib.sxfir_synth_fill(dev, 16, 16, 0, 1, 0, 0, 0, None), EINVAL)            # no channels
    # ordinary (pageable) host memory has no device pointer: unsupported, said so; page-locked memory has one
    host = np.zeros(1024, dtype=np.complex64)
    got = vp()
    rc = lib.sxfir_host_device_pointer(vp(host.ctypes.data), host.nbytes, C.byref(got))
    assert rc == EUNSUPPORTED and not got.value                 # (an answer, not a failure: no message is set for it)
    sxxcvr_amd.pin_array(host)
    try:
        assert lib.sxfir_host_device_pointer(vp(host.ctypes.data), host.nbytes, C.byref(got)) == 0 and got.value
    finally:
        sxxcvr_amd.unpin_array(host)
    assert lib.sxfir_free(dev) == 0
    # timing entry points
    ms = C.c_float()
    failed(lib, lib.sxfir_time_decimate(None, None, 0, 0, None, 0, 1, None, C.byref(ms)), EINVAL)
    plan = sxxcvr_amd.Resampler(DECIMATE, sxxcvr_amd.design_lowpass(128, 4), 4)
    failed(lib, lib.sxfir_time_decimate(plan._plan, None, 0, 0, None, 0, 0, None, C.byref(ms)), EINVAL)       # iters < 1
