"""ctypes loader for the in-tree native libraries.  Fails loudly when a
library is missing: the product has no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(_HERE, "lib")

_sxfir = None
_sxfir_prof = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sxfir error %d: %s" % (code, msg))
        self.code = code


def _share_hip_runtime_with_torch():
    # torch ships its own libamdhip64.so.7; importing it first makes our
    # library resolve the same runtime instance instead of a second copy.
    if os.environ.get("SXFIR_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load_sxfir(profiling=False):
    """Load sxxcvr_amd/lib/libsxfir.so and declare its prototypes.

    profiling=True loads libsxfir_prof.so instead: the same C ABI plus include/sxfir_prof.h, with the kernel
    A/B variants and environment knobs (tools/ and tests/test_gpu_variants.py; an explicit argument, never
    an environment switch: the product's arithmetic does not depend on the environment)."""
    global _sxfir, _sxfir_prof
    if profiling:
        if _sxfir_prof is not None:
            return _sxfir_prof
    elif _sxfir is not None:
        return _sxfir
    path = os.path.join(LIBDIR, "libsxfir_prof.so" if profiling else "libsxfir.so")
    if profiling and os.environ.get("SXFIR_PROF_LIB"):
        # tools only (tools/prev_lib.sh): the PROFILING library of another commit, built into sxxcvr_amd/lib/prev/, for
        # before / after timings on the same box.  Only a library in that directory is accepted, and only by the profiling
        # loader: the product library's path is never taken from the environment.
        cand = os.path.realpath(os.environ["SXFIR_PROF_LIB"])
        prev = os.path.realpath(os.path.join(LIBDIR, "prev"))
        if os.path.dirname(cand) != prev:
            raise ImportError("SXFIR_PROF_LIB must name a library under %s (tools/prev_lib.sh builds it there), got %s" % (prev, cand))
        path = cand
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build the HIP extension first (python -m sxxcvr_amd.build). "
            "There is no CPU fallback." % path)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path, mode=C.RTLD_LOCAL if profiling else C.RTLD_GLOBAL)
    vp, sz, i64, u64, ll, dbl, ci = C.c_void_p, C.c_size_t, C.c_int64, C.c_uint64, C.c_longlong, C.c_double, C.c_int
    P = C.POINTER
    sig = {
        "sxfir_abi_version": (ci, []),
        "sxfir_last_error": (C.c_char_p, []),
        "sxfir_device_count": (ci, [P(ci)]),
        "sxfir_device_info": (ci, [ci, C.c_char_p, C.c_char_p, P(ci), P(sz)]),
        "sxfir_device_pci_bus_id": (ci, [ci, C.c_char_p, sz]),
        "sxfir_create": (ci, [P(vp), ci, vp, ci, ci, ci, ci, ci]),
        "sxfir_destroy": (ci, [vp]),
        "sxfir_reset": (ci, [vp, vp]),
        "sxfir_set_history": (ci, [vp, vp, sz, sz, vp]),
        "sxfir_set_position": (ci, [vp, i64]),
        "sxfir_set_kernel": (ci, [vp, ci]),
        "sxfir_set_tx_threshold": (ci, [vp, C.c_float]),
        "sxfir_contract": (ci, [vp, P(ci), P(ci)]),
        "sxfir_contract_rotation": (ci, [vp, P(ci)]),
        "sxfir_position": (ci, [vp, P(i64), P(i64)]),
        "sxfir_outputs_for": (ci, [vp, sz, P(sz)]),
        "sxfir_launch_geometry": (ci, [vp, sz, vp]),
        "sxfir_decimate": (ci, [vp, vp, sz, sz, vp, sz, P(sz), vp]),
        "sxfir_interpolate": (ci, [vp, vp, sz, sz, vp, sz, P(sz), vp]),
        "sxfir_comm_unique_id": (ci, [vp]),
        "sxfir_comm_init_rank": (ci, [P(vp), vp, ci, ci, ci]),
        "sxfir_comm_init_all": (ci, [P(vp), ci, P(ci)]),
        "sxfir_comm_destroy": (ci, [vp]),
        "sxfir_comm_rank": (ci, [vp, P(ci), P(ci), P(ci)]),
        "sxfir_comm_query": (ci, [vp, P(ci), P(ci), P(ci)]),
        "sxfir_comm_gather": (ci, [vp, vp, vp, sz, sz, ci, sz, vp]),
        "sxfir_comm_gather_all": (ci, [P(vp), ci, P(vp), vp, sz, sz, ci, sz, P(vp)]),
        "sxfir_interpolate_keyed": (ci, [vp, vp, sz, sz, vp, sz, P(sz), sz, sz, vp, vp]),
        "sxfir_time_decimate": (ci, [vp, vp, sz, sz, vp, sz, ci, vp, P(C.c_float)]),
        "sxfir_time_interpolate": (ci, [vp, vp, sz, sz, vp, sz, ci, vp, P(C.c_float)]),
        "sxfir_clock_probe_start": (ci, [P(vp), ci, ci]),
        "sxfir_clock_probe_read": (ci, [vp, P(dbl)]),
        "sxfir_synth_fill": (ci, [vp, sz, sz, ci, u64, C.c_uint32, i64, ci, vp]),
        "sxfir_convert_rx_s32": (ci, [vp, vp, sz, vp]),
        "sxfir_convert_tx_s32": (ci, [vp, vp, sz, C.c_float, vp]),
        "sxfir_count_keyed": (ci, [vp, sz, C.c_float, vp, vp]),
        "sxfir_cf32_to_cf16": (ci, [vp, vp, sz, vp]),
        "sxfir_cf16_to_cf32": (ci, [vp, vp, sz, vp]),
        "sxfir_ticks_to_time_ns": (ll, [ll, dbl]),
        "sxfir_time_ns_to_ticks": (ll, [ll, dbl]),
        "sxfir_design_lowpass": (ci, [ci, ci, dbl, dbl, vp]),
        "sxfir_malloc": (ci, [P(vp), sz]),
        "sxfir_free": (ci, [vp]),
        "sxfir_set_device": (ci, [ci]),
        "sxfir_host_alloc": (ci, [P(vp), sz]),
        "sxfir_host_free": (ci, [vp]),
        "sxfir_host_register": (ci, [vp, sz]),
        "sxfir_host_unregister": (ci, [vp]),
        "sxfir_host_device_pointer": (ci, [vp, sz, P(vp)]),
        "sxfir_stream_create": (ci, [P(vp)]),
        "sxfir_stream_destroy": (ci, [vp]),
        "sxfir_event_create": (ci, [P(vp)]),
        "sxfir_event_destroy": (ci, [vp]),
        "sxfir_event_create_timing": (ci, [P(vp)]),
        "sxfir_event_elapsed_ms": (ci, [vp, vp, P(C.c_float)]),
        "sxfir_event_record": (ci, [vp, vp]),
        "sxfir_event_sync": (ci, [vp]),
        "sxfir_stream_wait_event": (ci, [vp, vp]),
        "sxfir_memcpy_h2d": (ci, [vp, vp, sz, vp]),
        "sxfir_memcpy_d2h": (ci, [vp, vp, sz, vp]),
        "sxfir_stream_sync": (ci, [vp]),
    }
    if profiling:
        sig["sxfir_debug_clock"] = (ci, [vp, P(dbl)])
        sig["sxfir_debug_stamps"] = (ci, [vp, P(C.c_ulonglong), sz, P(sz)])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    lib._sx_signatures = sig
    if profiling:
        _sxfir_prof = lib
    else:
        _sxfir = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        raise NativeError(rc, (lib or load_sxfir()).sxfir_last_error().decode("utf-8", "replace"))
