"""ctypes binding of the CPU oracle (oracle/libsxoracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SXO_ORACLE_DIR: a copy of oracle/ to build and load instead (the build-race test works on a copy under tmp_path,
# so that it never touches the tracked sources or swaps a library under a concurrently running session)
_ODIR = os.environ.get("SXO_ORACLE_DIR") or os.path.join(_ROOT, "oracle")


class StreamResult(C.Structure):
    _fields_ = [
        ("position", C.c_int64),
        ("skipped", C.c_int64),
        ("length", C.c_int64),
        ("time_ns", C.c_longlong),
        ("flags", C.c_int),
        ("ret", C.c_int),
        ("discarded", C.c_int),
    ]


def build():
    """Compile the oracle if a library is missing or older than its source.  The check and the `make` are one
    critical section under an flock: the ranks of `bench.py --gpus N` all come through here at once."""
    import fcntl
    src = os.path.join(_ODIR, "sx_oracle.c")
    hdr = os.path.join(_ODIR, "sx_oracle.h")
    with open(os.path.join(_ODIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            newest = max(os.path.getmtime(src), os.path.getmtime(hdr))
            for name in ("libsxoracle.so", "libsxoracle_fast.so"):
                p = os.path.join(_ODIR, name)
                if not os.path.exists(p) or os.path.getmtime(p) < newest:
                    subprocess.check_call(["make", "-C", _ODIR, "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    break
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def _load(name):
    path = os.path.join(_ODIR, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, i64, u64, sz, dbl, ll = C.c_void_p, C.c_int64, C.c_uint64, C.c_size_t, C.c_double, C.c_longlong
    lib.sxo_ticks_to_time_ns.restype = ll
    lib.sxo_ticks_to_time_ns.argtypes = [ll, dbl]
    lib.sxo_time_ns_to_ticks.restype = ll
    lib.sxo_time_ns_to_ticks.argtypes = [ll, dbl]
    lib.sxo_convert_rx.argtypes = [vp, vp, sz]
    lib.sxo_convert_rx_mt.argtypes = [vp, vp, sz, C.c_int]
    lib.sxo_synth_iq_mt.argtypes = [u64, C.c_uint32, i64, sz, vp, C.c_int]
    lib.sxo_interp_f32_mt.restype = C.c_int
    lib.sxo_interp_f32_mt.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp, C.c_int]
    lib.sxo_convert_tx.argtypes = [vp, vp, sz, C.c_float]
    lib.sxo_synth_iq.argtypes = [u64, C.c_uint32, i64, sz, vp]
    lib.sxo_design_lowpass.argtypes = [C.c_int, C.c_int, dbl, dbl, vp]
    for f in (lib.sxo_decim_f64, lib.sxo_interp_f64):
        f.restype = C.c_int
        f.argtypes = [vp, C.c_int, C.c_int, vp, sz, i64, sz, vp]
    lib.sxo_interp_f32.restype = C.c_int
    lib.sxo_interp_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp]
    lib.sxo_decim_f32.restype = C.c_int
    lib.sxo_decim_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp]
    lib.sxo_decim_f32_mt.restype = C.c_int
    lib.sxo_decim_f32_mt.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp, C.c_int]
    lib.sxo_decim_f32_rot.restype = C.c_int
    lib.sxo_decim_f32_rot.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp]
    lib.sxo_decim_f32_rot_mt.restype = C.c_int
    lib.sxo_decim_f32_rot_mt.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, i64, sz, vp, C.c_int]
    lib.sxo_max_threads.restype = C.c_int
    lib.sxo_f32_to_f16.argtypes = [vp, vp, sz]
    lib.sxo_f16_to_f32.argtypes = [vp, vp, sz]
    lib.sxo_gain_split.argtypes = [C.c_int, dbl, C.POINTER(dbl), C.POINTER(dbl)]
    lib.sxo_quantize_frequency.restype = dbl
    lib.sxo_quantize_frequency.argtypes = [dbl, dbl, C.POINTER(C.c_uint)]
    lib.sxo_rx_step.argtypes = [i64, i64, u64, u64, sz, C.c_long, dbl, C.POINTER(StreamResult)]
    lib.sxo_tx_step.argtypes = [i64, i64, i64, u64, sz, C.c_int, ll, C.c_long, dbl, C.POINTER(StreamResult)]
    return lib


class Oracle:
    """numpy-facing view of the C oracle.  Complex data = complex64 arrays."""

    def __init__(self, fast=False):
        self.lib = _load("libsxoracle_fast.so" if fast else "libsxoracle.so")

    # -- time ---------------------------------------------------------------
    def ticks_to_time_ns(self, ticks, rate):
        return int(self.lib.sxo_ticks_to_time_ns(int(ticks), float(rate)))

    def time_ns_to_ticks(self, ns, rate):
        return int(self.lib.sxo_time_ns_to_ticks(int(ns), float(rate)))

    # -- conversion ---------------------------------------------------------
    def convert_rx(self, s32):
        s32 = np.ascontiguousarray(s32, dtype=np.int32)
        n = s32.size // 2
        out = np.empty(n, dtype=np.complex64)
        self.lib.sxo_convert_rx(_fp(s32), _fp(out), n)
        return out

    def convert_rx_into(self, s32, out, threads=1):
        """convert_rx over `threads` threads into a caller-owned float32 array (timing aid: no allocation)."""
        self.lib.sxo_convert_rx_mt(_fp(s32), _fp(out), s32.size // 2, int(threads))
        return out

    def convert_tx(self, cf32, threshold2):
        cf32 = np.ascontiguousarray(cf32, dtype=np.complex64)
        out = np.empty(2 * cf32.size, dtype=np.int32)
        self.lib.sxo_convert_tx(_fp(cf32), _fp(out), cf32.size, np.float32(threshold2))
        return out

    # -- source / taps ------------------------------------------------------
    def synth_iq(self, seed, channel, start, n):
        out = np.empty(n, dtype=np.complex64)
        self.lib.sxo_synth_iq(int(seed), int(channel), int(start), n, _fp(out))
        return out

    def synth_iq_mt(self, seed, channel, start, n, threads):
        out = np.empty(n, dtype=np.complex64)
        self.lib.sxo_synth_iq_mt(int(seed), int(channel), int(start), n, _fp(out), int(threads))
        return out

    def design_lowpass(self, ntaps, ratio, beta=8.0, gain=1.0):
        out = np.empty(ntaps, dtype=np.float32)
        self.lib.sxo_design_lowpass(ntaps, ratio, float(beta), float(gain), _fp(out))
        return out

    # -- FIR ----------------------------------------------------------------
    def _run(self, fn, h, ratio, x, o0, n_out, groups=None, threads=None):
        h = np.ascontiguousarray(h, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.complex64)
        y = np.empty(n_out, dtype=np.complex64)
        args = [_fp(h), h.size, int(ratio)]
        if groups is not None:
            args += [int(g) for g in (groups if isinstance(groups, tuple) else (groups,))]
        args += [_fp(x), x.size, int(o0), n_out, _fp(y)]
        if threads is not None:
            args.append(int(threads))
        rc = fn(*args)
        if rc != 0:
            raise ValueError("oracle returned %d" % rc)
        return y

    def decim_f64(self, h, D, x, m0=0, n_out=None):
        n_out = (len(x) + D - 1) // D - m0 if n_out is None else n_out
        return self._run(self.lib.sxo_decim_f64, h, D, x, m0, n_out)

    def decim_f32(self, h, D, x, jsplit=1, cw=None, m0=0, n_out=None, threads=None, rot=0):
        """Order-matched fp32 decimator; (jsplit, cw) and the rotation (Resampler.contract.rot) are the kernel's contract."""
        n_out = (len(x) + D - 1) // D - m0 if n_out is None else n_out
        g = (jsplit, D if cw is None else cw, rot)
        if threads is None:
            return self._run(self.lib.sxo_decim_f32_rot, h, D, x, m0, n_out, groups=g)
        return self._run(self.lib.sxo_decim_f32_rot_mt, h, D, x, m0, n_out, groups=g, threads=threads)

    def interp_f64(self, h, L, x, n0=0, n_out=None):
        n_out = len(x) * L - n0 if n_out is None else n_out
        return self._run(self.lib.sxo_interp_f64, h, L, x, n0, n_out)

    def interp_f32(self, h, L, x, groups, n0=0, n_out=None):
        n_out = len(x) * L - n0 if n_out is None else n_out
        return self._run(self.lib.sxo_interp_f32, h, L, x, n0, n_out, groups=groups)

    def interp_f32_mt(self, h, L, x, groups, n0=0, n_out=None, threads=1):
        n_out = len(x) * L - n0 if n_out is None else n_out
        return self._run(self.lib.sxo_interp_f32_mt, h, L, x, n0, n_out, groups=groups, threads=threads)

    def max_threads(self):
        """Threads worth starting: OpenMP's own figure, capped by the CPUs this process may use (affinity mask)
        and by the cgroup CPU quota (a container with cpu.max = 16 CPUs on a 256-thread host runs 16 threads well
        and 128 threads badly)."""
        n = int(self.lib.sxo_max_threads())
        try:
            n = min(n, len(os.sched_getaffinity(0)))
        except (AttributeError, OSError):
            pass
        for path in ("/sys/fs/cgroup/cpu.max",):
            try:
                quota, period = open(path).read().split()[:2]
                if quota != "max":
                    n = min(n, max(1, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
        return max(1, n)

    # -- half ---------------------------------------------------------------
    def f32_to_f16(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty(a.shape, dtype=np.uint16)
        self.lib.sxo_f32_to_f16(_fp(a), _fp(out), a.size)
        return out

    def f16_to_f32(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint16)
        out = np.empty(a.shape, dtype=np.float32)
        self.lib.sxo_f16_to_f32(_fp(a), _fp(out), a.size)
        return out

    # -- control surface ----------------------------------------------------
    def gain_split(self, direction, value):
        a, b = C.c_double(), C.c_double()
        self.lib.sxo_gain_split(direction, float(value), C.byref(a), C.byref(b))
        return a.value, b.value

    def quantize_frequency(self, master_clock, frequency):
        w = C.c_uint()
        f = self.lib.sxo_quantize_frequency(float(master_clock), float(frequency), C.byref(w))
        return f, w.value

    # -- stream rules -------------------------------------------------------
    def rx_step(self, position, avail, period, buffer, num_elems, timeout_us, rate):
        r = StreamResult()
        self.lib.sxo_rx_step(position, avail, period, buffer, num_elems, timeout_us, rate, C.byref(r))
        return r

    def tx_step(self, position, avail, delay, period, num_elems, flags, time_ns, timeout_us, rate):
        r = StreamResult()
        self.lib.sxo_tx_step(position, avail, delay, period, num_elems, flags, time_ns, timeout_us, rate, C.byref(r))
        return r
