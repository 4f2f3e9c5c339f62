"""Python face of the C ABI in include/sxfir.h.

Device memory comes either from torch (complex64 / int32 CUDA tensors, the
current torch stream is used) or from the library's own sxfir_malloc for
numpy-only callers (``*_host`` helpers).  No arithmetic happens in Python.
"""
import ctypes as C

import numpy as np

from ._native import check as _check, load_sxfir

DECIMATE, INTERPOLATE = 0, 1
CF32, CF16, S32 = 0, 1, 2
KERNEL_AUTO, KERNEL_TILED, KERNEL_GENERIC = 0, 1, 2
_FMT = {"CF32": CF32, "CF16": CF16, "S32": S32, CF32: CF32, CF16: CF16, S32: S32}


class Contract(tuple):
    """(jsplit, cw) with the rotation beside it: unpacks and compares as the pair it always was.  Under a rotation
    (rot != 0: the decimators by 48 and 96) the pair ALONE does not state the summation order -- unpacking it without ever
    having read .rot warns once (a caller written before ABI 5 would feed (2, 4) to its own check and get other bits)."""
    def __new__(cls, pair, rot=0):
        self = super().__new__(cls, pair)
        self._rot = rot
        self._rot_seen = rot == 0
        return self

    @property
    def rot(self):
        self._rot_seen = True
        return self._rot

    def __iter__(self):
        if not self._rot_seen:
            import warnings
            self._rot_seen = True
            warnings.warn("this plan's contract (%d, %d) holds under rotation %d (sxfir_contract_rotation): read Contract.rot too"
                          % (self[0], self[1], self._rot), stacklevel=2)
        return super().__iter__()


def check(rc, lib=None):
    _check(rc, lib)


def design_lowpass(ntaps, ratio, beta=8.0, gain=1.0):
    lib = load_sxfir()
    taps = np.empty(ntaps, dtype=np.float32)
    check(lib.sxfir_design_lowpass(ntaps, ratio, beta, gain, taps.ctypes.data_as(C.c_void_p)))
    return taps


def _torch_stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def synth_fill(out, seed, first_channel=0, start=0, fmt="CF32"):
    """Fill a CUDA tensor [nchan, n] (complex64 for CF32, int32 words for CF16)
    with the synthetic IQ source."""
    lib = load_sxfir()
    t = out if out.dim() == 2 else out.unsqueeze(0)
    nchan, n = t.shape
    check(lib.sxfir_synth_fill(C.c_void_p(t.data_ptr()), n, t.stride(0), nchan, seed, first_channel, start,
                               _FMT[fmt], _torch_stream(t)))
    return out


def pin_array(a):
    """Page-lock a numpy array's memory (sxfir_host_register) so that the GPU can store into it directly: large
    readStream calls into such a buffer skip the staging copy.  Undo with unpin_array before the array is freed."""
    lib = load_sxfir()
    check(lib.sxfir_host_register(C.c_void_p(a.ctypes.data), a.nbytes))
    return a


def unpin_array(a):
    check(load_sxfir().sxfir_host_unregister(C.c_void_p(a.ctypes.data)))


class StreamTimer:
    """GPU time of whatever is launched on `stream` between start() and stop(): HIP events recorded on that
    stream through the C ABI (torch.cuda.Event would only see torch's current stream)."""

    def __init__(self, stream=0):
        self._lib = load_sxfir()
        self._stream = C.c_void_p(stream)
        self._e0, self._e1 = C.c_void_p(), C.c_void_p()
        check(self._lib.sxfir_event_create_timing(C.byref(self._e0)))
        check(self._lib.sxfir_event_create_timing(C.byref(self._e1)))

    def start(self):
        check(self._lib.sxfir_event_record(self._e0, self._stream))

    def stop(self):
        check(self._lib.sxfir_event_record(self._e1, self._stream))

    def elapsed_ms(self):
        ms = C.c_float()
        check(self._lib.sxfir_event_elapsed_ms(self._e0, self._e1, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            self._lib.sxfir_event_destroy(self._e0)
            self._lib.sxfir_event_destroy(self._e1)
        except Exception:
            pass


class ClockProbe:
    """In-kernel shader clock while other work runs (sxfir_clock_probe_*): start, run the work, read()."""

    def __init__(self, device=-1, duration_us=20000):
        self._lib = load_sxfir()
        self._h = C.c_void_p()
        check(self._lib.sxfir_clock_probe_start(C.byref(self._h), int(device), int(duration_us)))

    def read(self):
        mhz = C.c_double()
        h, self._h = self._h, C.c_void_p()
        check(self._lib.sxfir_clock_probe_read(h, C.byref(mhz)))
        return mhz.value


class Geometry(C.Structure):
    """sxfir_geometry of include/sxfir.h: what a call would launch."""
    _fields_ = [("kernel", C.c_char * 64), ("tiled", C.c_int), ("split", C.c_int), ("tile_samples", C.c_longlong),
                ("n_tiles", C.c_longlong), ("workgroups", C.c_longlong), ("resident", C.c_longlong)]


class Resampler:
    """One sxfir plan: `nchan` independent channels of one GPU."""

    def __init__(self, mode, taps, ratio, nchan=1, fmt="CF32", device=-1, profiling=False):
        self._lib = load_sxfir(profiling)
        self._plan = C.c_void_p()
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.mode, self.ratio, self.nchan, self.fmt, self.ntaps = mode, int(ratio), int(nchan), _FMT[fmt], taps.size
        self._ck(self._lib.sxfir_create(C.byref(self._plan), mode, taps.ctypes.data_as(C.c_void_p), taps.size,
                                     int(ratio), int(nchan), self.fmt, int(device)))

    def _ck(self, rc):
        _check(rc, self._lib)      # error text from the library this plan lives in

    def close(self):
        if self._plan:
            self._lib.sxfir_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection --------------------------------------------------------
    @property
    def contract(self):
        """(jsplit, cw) of the plan's numeric contract; the tuple's attribute `rot` is the contract's rotation
        (sxfir_contract_rotation: 1 for the decimators by 48 and 96, else 0)."""
        a, b, r = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._lib.sxfir_contract(self._plan, C.byref(a), C.byref(b)))
        self._ck(self._lib.sxfir_contract_rotation(self._plan, C.byref(r)))
        return Contract((a.value, b.value), r.value)

    @property
    def position(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self._lib.sxfir_position(self._plan, C.byref(a), C.byref(b)))
        return a.value, b.value

    def geometry(self, n_in):
        """What a call with n_in new input samples would launch now (sxfir_launch_geometry): kernel family, whether an
        LDS-tiled kernel runs it, tiles, work items per tile, workgroups, and the workgroup slots of the chip."""
        g = Geometry()
        self._ck(self._lib.sxfir_launch_geometry(self._plan, n_in, C.byref(g)))
        return {"kernel": g.kernel.decode(), "tiled": bool(g.tiled), "split": g.split, "tile_samples": g.tile_samples,
                "n_tiles": g.n_tiles, "workgroups": g.workgroups, "resident": g.resident}

    def outputs_for(self, n_in):
        n = C.c_size_t()
        self._ck(self._lib.sxfir_outputs_for(self._plan, n_in, C.byref(n)))
        return n.value

    def set_kernel(self, kernel):
        self._ck(self._lib.sxfir_set_kernel(self._plan, kernel))

    def set_tx_threshold(self, threshold2):
        self._ck(self._lib.sxfir_set_tx_threshold(self._plan, float(threshold2)))

    def reset(self, stream=None):
        self._ck(self._lib.sxfir_reset(self._plan, C.c_void_p(stream or 0)))

    def set_history_ptr(self, src_ptr, n, stride, stream=0):
        """The filter state becomes the last samples of the device block [src_ptr, src_ptr + n) of every channel
        (sxfir_set_history): as if that block had just been processed."""
        self._ck(self._lib.sxfir_set_history(self._plan, C.c_void_p(src_ptr), n, stride, C.c_void_p(stream)))

    def set_position(self, consumed):
        """Place the plan at input sample `consumed` of its stream (sxfir_set_position): a decimator's output phase
        and output count of the next call follow from it."""
        self._ck(self._lib.sxfir_set_position(self._plan, int(consumed)))

    # -- raw pointers ---------------------------------------------------------
    def process_ptr(self, in_ptr, n_in, in_stride, out_ptr, out_stride, stream=0):
        n_out = C.c_size_t()
        fn = self._lib.sxfir_decimate if self.mode == DECIMATE else self._lib.sxfir_interpolate
        self._ck(fn(self._plan, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr), out_stride, C.byref(n_out),
                 C.c_void_p(stream)))
        return n_out.value

    def interpolate_keyed_ptr(self, in_ptr, n_in, in_stride, out_ptr, out_stride, key_first, key_count, counter_ptr,
                              stream=0):
        """sxfir_interpolate_keyed: the interpolation pass that also adds to the 8-byte device word at counter_ptr how
        many of channel 0's input samples [key_first, key_first + key_count) reach the keying threshold
        (convert_tx_buffer, SoapySX.cpp:132-133)."""
        n_out = C.c_size_t()
        self._ck(self._lib.sxfir_interpolate_keyed(self._plan, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr),
                                                   out_stride, C.byref(n_out), key_first, key_count,
                                                   C.c_void_p(counter_ptr), C.c_void_p(stream)))
        return n_out.value

    def time_decimate_ptr(self, in_ptr, n_in, in_stride, out_ptr, out_stride, iters, stream=0):
        return self.time_passes_ptr(in_ptr, n_in, in_stride, out_ptr, out_stride, iters, stream)

    def time_passes_ptr(self, in_ptr, n_in, in_stride, out_ptr, out_stride, iters, stream=0):
        """Mean milliseconds of `iters` back-to-back launches of the resampling kernel (HIP events on `stream`);
        the filter state is left alone."""
        ms = C.c_float()
        fn = self._lib.sxfir_time_decimate if self.mode == DECIMATE else self._lib.sxfir_time_interpolate
        self._ck(fn(self._plan, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr), out_stride, iters,
                    C.c_void_p(stream), C.byref(ms)))
        return ms.value

    # -- torch tensors --------------------------------------------------------
    def process(self, x, out=None):
        """x: CUDA tensor [nchan, n] or [n]; complex64 (CF32) or int32 words (CF16).
        S32 plans: a decimator takes int32 wire words [.., n, 2] (I, Q) and returns complex64; an
        interpolator takes complex64 and returns int32 wire words [.., n_out, 2]."""
        import torch
        if self.fmt == S32:
            if self.mode == DECIMATE:
                xc = torch.view_as_complex(x.view(torch.float32))          # same 8 bytes per sample
                return self._process(xc, out, torch.complex64)
            y = self._process(x, None if out is None else torch.view_as_complex(out.view(torch.float32)),
                              torch.complex64)
            return torch.view_as_real(y).view(torch.int32)
        return self._process(x, out, x.dtype)

    def _process(self, x, out, out_dtype):
        import torch
        squeeze = x.dim() == 1
        x2 = x.unsqueeze(0) if squeeze else x
        if x2.shape[0] != self.nchan or x2.stride(1) != 1:
            raise ValueError("expected a [nchan=%d, n] tensor with unit sample stride" % self.nchan)
        n_in = x2.shape[1]
        n_out = self.outputs_for(n_in)
        if out is None:
            out = torch.empty((self.nchan, n_out), dtype=out_dtype, device=x2.device)
        o2 = out.unsqueeze(0) if out.dim() == 1 else out
        # the kernel writes through a raw pointer: a caller-supplied `out` must really hold the result
        if (o2.dim() != 2 or o2.shape[0] != self.nchan or o2.shape[1] < n_out or (o2.shape[1] > 1 and o2.stride(1) != 1)
                or o2.dtype != out_dtype or o2.device != x2.device):
            raise ValueError("out must be a [nchan=%d, >=%d] %s tensor with unit sample stride on %s" % (
                self.nchan, n_out, out_dtype, x2.device))
        got = self.process_ptr(x2.data_ptr(), n_in, x2.stride(0) if self.nchan > 1 else n_in, o2.data_ptr(),
                               o2.stride(0) if self.nchan > 1 else max(n_out, 1),
                               torch.cuda.current_stream(x2.device).cuda_stream)
        assert got == n_out
        res = o2[:, :n_out]
        return res[0] if squeeze else res

    # -- numpy host arrays (library-owned device memory) ----------------------
    def process_host(self, x):
        """x: complex64 numpy [nchan, n] or [n] (CF32 plans only).  Copies to the
        GPU, runs the HIP path, copies back."""
        if self.fmt != CF32:
            raise ValueError("process_host handles CF32 plans")
        lib = self._lib
        x = np.ascontiguousarray(x, dtype=np.complex64)
        squeeze = x.ndim == 1
        x2 = x.reshape(1, -1) if squeeze else x
        nchan, n_in = x2.shape
        n_out = self.outputs_for(n_in)
        y = np.empty((nchan, n_out), dtype=np.complex64)
        din, dout = C.c_void_p(), C.c_void_p()
        check(lib.sxfir_malloc(C.byref(din), max(x2.nbytes, 16)))
        check(lib.sxfir_malloc(C.byref(dout), max(y.nbytes, 16)))
        try:
            if x2.nbytes:
                check(lib.sxfir_memcpy_h2d(din, x2.ctypes.data_as(C.c_void_p), x2.nbytes, None))
            self.process_ptr(din.value, n_in, n_in, dout.value, max(n_out, 1))
            if y.nbytes:
                check(lib.sxfir_memcpy_d2h(y.ctypes.data_as(C.c_void_p), dout, y.nbytes, None))
            check(lib.sxfir_stream_sync(None))
        finally:
            lib.sxfir_free(din)
            lib.sxfir_free(dout)
        return y[0] if squeeze else y


class PipelinedResampler:
    """Consecutive blocks of ONE stream on `depth` plans and `depth` HIP streams in turn, so that their passes
    overlap on the GPU.  Block k+1 needs nothing of block k but the tail of its INPUT, which is in device memory
    before either pass runs: each plan's filter state is seeded from there (sxfir_set_history) and the passes are
    independent launches.  Each plan is also told where in the stream its block starts (sxfir_set_position), so
    blocks need not be multiples of the ratio: the outputs, their count and their positions are those of one plan
    fed block by block.  (On one GPU this buys nothing once every block is new data -- LABBOOK.md 7 -- it is the way
    to split one stream over plans or GPUs.)

    Ordering: every pass waits (on the GPU) for the work the CALLER's current torch stream had queued when
    process_ptr was called -- the kernels that produced the block -- and nothing else.  The caller keeps every input
    block unchanged until the pass over the NEXT block has run (join() waits for all)."""

    def __init__(self, mode, taps, ratio, nchan=1, fmt="CF32", device=-1, depth=4):
        import torch
        self.depth = int(depth)
        self.plans = [Resampler(mode, taps, ratio, nchan=nchan, fmt=fmt, device=device) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream(device=None if device < 0 else device) for _ in range(self.depth)]
        self._k = 0
        self._prev = None                                   # (ptr, n, stride) of the previous input block
        self._consumed = 0                                  # input samples of the stream before the next block

    @property
    def contract(self):
        return self.plans[0].contract

    def restart(self):
        """The next block is the first of a new stream (zero history, position 0)."""
        self._prev = None
        self._consumed = 0

    def process_ptr(self, in_ptr, n_in, in_stride, out_ptr, out_stride):
        """Queue the pass over one block; returns its output count.  Asynchronous: join() before using outputs."""
        import torch
        k = self._k % self.depth
        plan, st = self.plans[k], self.streams[k].cuda_stream
        # the block (and the previous one, whose tail is the history) was produced on the caller's stream
        self.streams[k].wait_stream(torch.cuda.current_stream(self.streams[k].device))
        if self._prev is None:
            plan.reset(st)
        else:
            # (a previous block shorter than the filter history is refused by sxfir_set_history: SXFIR_EINVAL)
            plan.set_history_ptr(self._prev[0], self._prev[1], self._prev[2], st)
        plan.set_position(self._consumed)
        n_out = plan.process_ptr(in_ptr, n_in, in_stride, out_ptr, out_stride, st)
        self._prev = (in_ptr, n_in, in_stride)
        self._consumed += n_in
        self._k += 1
        return n_out

    def join(self):
        for s in self.streams:
            s.synchronize()
