"""Drop-in stand-in for the ``SoapySDR`` Python module, for the calls that
tejeez/sxxcvr's scripts make (example/*.py, SoapySX/test/*.py):

    import sxxcvr_amd.soapy as SoapySDR
    dev = SoapySDR.Device({'driver': 'sx'})
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {})
    r = dev.readStream(rx, [buf], len(buf))        # r.ret, r.flags, r.timeNs

It binds the flat C view (include/sx_device.h) of the C++ SoapySDR::Device
module in sxxcvr_amd/lib/libSXSupport.so with ctypes.  With a real SoapySDR
installation the same module is loaded by SoapySDR itself and this file is
not needed.
"""
import ctypes as C
import os

import numpy as np

from . import _native

SOAPY_SDR_TX, SOAPY_SDR_RX = 0, 1
SOAPY_SDR_END_BURST, SOAPY_SDR_HAS_TIME = 2, 4
SOAPY_SDR_TIMEOUT, SOAPY_SDR_STREAM_ERROR, SOAPY_SDR_CORRUPTION = -1, -2, -3
SOAPY_SDR_OVERFLOW, SOAPY_SDR_NOT_SUPPORTED, SOAPY_SDR_TIME_ERROR, SOAPY_SDR_UNDERFLOW = -4, -5, -6, -7
SOAPY_SDR_CF32 = "CF32"


class Range:
    """SoapySDR::Range as the Python bindings present it."""

    def __init__(self, minimum, maximum, step=0.0):
        self._min, self._max, self._step = float(minimum), float(maximum), float(step)

    def minimum(self):
        return self._min

    def maximum(self):
        return self._max

    def step(self):
        return self._step

    def __repr__(self):
        return "%g, %g, %g" % (self._min, self._max, self._step)

SOAPY_SDR_FATAL, SOAPY_SDR_CRITICAL, SOAPY_SDR_ERROR, SOAPY_SDR_WARNING = 1, 2, 3, 4
SOAPY_SDR_NOTICE, SOAPY_SDR_INFO, SOAPY_SDR_DEBUG, SOAPY_SDR_TRACE = 5, 6, 7, 8

_EXC = -1000
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    _native.load_sxfir()                       # dependency, loaded RTLD_GLOBAL first
    path = os.path.join(_native.LIBDIR, "libSXSupport.so")
    if not os.path.exists(path):
        raise ImportError("%s is missing: run python -m sxxcvr_amd.build" % path)
    lib = C.CDLL(path)
    vp, sz, ll, dbl, ci, cl, cs = C.c_void_p, C.c_size_t, C.c_longlong, C.c_double, C.c_int, C.c_long, C.c_char_p
    P = C.POINTER
    sig = {
        "sx_device_last_error": (cs, []),
        "sx_device_enumerate": (ci, [cs, cs, sz]),
        "sx_device_make": (vp, [cs]),
        "sx_device_unmake": (ci, [vp]),
        "sx_device_setup_stream": (vp, [vp, ci, cs, P(sz), sz, cs]),
        "sx_device_close_stream": (ci, [vp, vp]),
        "sx_device_get_stream_mtu": (cl, [vp, vp]),
        "sx_device_activate_stream": (ci, [vp, vp, ci, ll, sz]),
        "sx_device_deactivate_stream": (ci, [vp, vp, ci, ll]),
        "sx_device_read_stream": (ci, [vp, vp, P(vp), sz, P(ci), P(ll), cl]),
        "sx_device_write_stream": (ci, [vp, vp, P(vp), sz, P(ci), ll, cl]),
        "sx_device_has_hardware_time": (ci, [vp, cs]),
        "sx_device_get_hardware_time": (ci, [vp, cs, P(ll)]),
        "sx_device_list_sample_rates": (ci, [vp, ci, sz, P(dbl), sz]),
        "sx_device_set_sample_rate": (ci, [vp, ci, sz, dbl]),
        "sx_device_get_sample_rate": (dbl, [vp, ci, sz]),
        "sx_device_get_num_channels": (ci, [vp, ci]),
        "sx_device_get_gain_range": (ci, [vp, ci, sz, C.c_char_p, P(C.c_double)]),
        "sx_device_get_info": (ci, [vp, cs, ci, cs, sz]),
        "sx_device_set_frequency": (ci, [vp, ci, sz, dbl]),
        "sx_device_get_frequency": (dbl, [vp, ci, sz]),
        "sx_device_set_gain": (ci, [vp, ci, sz, dbl]),
        "sx_device_get_gain": (dbl, [vp, ci, sz]),
        "sx_device_set_antenna": (ci, [vp, ci, sz, cs]),
        "sx_device_get_antenna": (ci, [vp, ci, sz, cs, sz]),
        "sx_device_set_gain_element": (ci, [vp, ci, sz, cs, dbl]),
        "sx_device_get_gain_element": (dbl, [vp, ci, sz, cs]),
        "sx_device_list": (ci, [vp, cs, ci, cs, sz]),
        "sx_device_write_registers": (ci, [vp, cs, C.c_uint, P(C.c_uint), sz]),
        "sx_device_read_registers": (ci, [vp, cs, C.c_uint, P(C.c_uint), sz]),
        "sx_device_write_setting": (ci, [vp, cs, cs]),
        "sx_device_read_setting": (ci, [vp, cs, cs, sz]),
        "sx_device_tx_capture": (ci, [vp, ll, sz, vp]),
        "sx_ticks_to_time_ns": (ll, [ll, dbl]),
        "sx_time_ns_to_ticks": (ll, [ll, dbl]),
        "sx_device_set_log_level": (ci, [ci]),
        "sx_device_drain_log": (ci, [cs, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._sx_signatures = sig
    _lib = lib
    return lib


def _err():
    return _load().sx_device_last_error().decode("utf-8", "replace")


def _kwargs(args):
    if args is None:
        return b""
    if isinstance(args, str):
        return args.encode()
    return ", ".join("%s=%s" % (k, v) for k, v in dict(args).items()).encode()


def _parse(markup):
    out = {}
    for item in markup.split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def ticksToTimeNs(ticks, rate):
    return int(_load().sx_ticks_to_time_ns(int(ticks), float(rate)))


def timeNsToTicks(time_ns, rate):
    return int(_load().sx_time_ns_to_ticks(int(time_ns), float(rate)))


def setLogLevel(level):
    _load().sx_device_set_log_level(int(level))


def drainLog():
    """Log lines emitted by the module since the last call (test helper)."""
    lib = _load()
    buf = C.create_string_buffer(1 << 20)
    lib.sx_device_drain_log(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


class StreamResult:
    def __init__(self, ret=0, flags=0, timeNs=0):
        self.ret, self.flags, self.timeNs, self.chanMask = ret, flags, timeNs, 0

    def __repr__(self):
        return "ret=%d, flags=%d, timeNs=%d" % (self.ret, self.flags, self.timeNs)


class Device:
    def __init__(self, args=None):
        self._lib = _load()
        self._dev = self._lib.sx_device_make(_kwargs(args))
        if not self._dev:
            raise RuntimeError(_err())

    @staticmethod
    def enumerate(args=None):
        lib = _load()
        buf = C.create_string_buffer(4096)
        n = lib.sx_device_enumerate(_kwargs(args), buf, len(buf))
        if n == _EXC:
            raise RuntimeError(_err())
        return [_parse(e) for e in buf.value.decode().split(";") if e][:n]

    def close(self):
        if getattr(self, "_dev", None):
            self._lib.sx_device_unmake(self._dev)
            self._dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc == _EXC:
            raise RuntimeError(_err())
        return rc

    def _str(self, fn, *a):
        buf = C.create_string_buffer(4096)
        self._chk(fn(self._dev, *a, buf, len(buf)))
        return buf.value.decode()

    # -- identification ----------------------------------------------------
    def getDriverKey(self):
        return self._str(self._lib.sx_device_get_info, b"driver_key", 0)

    def getHardwareKey(self):
        return self._str(self._lib.sx_device_get_info, b"hardware_key", 0)

    def getHardwareInfo(self):
        return _parse(self._str(self._lib.sx_device_get_info, b"hardware_info", 0))

    def getNumChannels(self, direction):
        return self._chk(self._lib.sx_device_get_num_channels(self._dev, direction))

    def getStreamFormats(self, direction, channel):
        return self._str(self._lib.sx_device_get_info, b"stream_formats", direction).split(",")

    def getNativeStreamFormat(self, direction, channel):
        fmt, fs = self._str(self._lib.sx_device_get_info, b"native_stream_format", direction).split(",")
        return fmt, float(fs)

    # -- rates, tuning -----------------------------------------------------
    def listSampleRates(self, direction, channel):
        arr = (C.c_double * 32)()
        n = self._chk(self._lib.sx_device_list_sample_rates(self._dev, direction, channel, arr, 32))
        return [arr[i] for i in range(n)]

    def setSampleRate(self, direction, channel, rate):
        self._chk(self._lib.sx_device_set_sample_rate(self._dev, direction, channel, float(rate)))

    def getSampleRate(self, direction, channel):
        return self._lib.sx_device_get_sample_rate(self._dev, direction, channel)

    def setFrequency(self, direction, channel, hz, args=None):
        self._chk(self._lib.sx_device_set_frequency(self._dev, direction, channel, float(hz)))

    def getFrequency(self, direction, channel):
        return self._lib.sx_device_get_frequency(self._dev, direction, channel)

    def setGain(self, direction, channel, *args):
        """setGain(dir, ch, value) or setGain(dir, ch, name, value), as in SoapySDR."""
        if len(args) == 1:
            self._chk(self._lib.sx_device_set_gain(self._dev, direction, channel, float(args[0])))
        else:
            self._chk(self._lib.sx_device_set_gain_element(self._dev, direction, channel, args[0].encode(),
                                                           float(args[1])))

    def getGain(self, direction, channel, name=None):
        if name is None:
            return self._lib.sx_device_get_gain(self._dev, direction, channel)
        return self._lib.sx_device_get_gain_element(self._dev, direction, channel, name.encode())

    def getGainRange(self, direction, channel, name=None):
        """Overall range, or one element's; the returned object has minimum() / maximum() / step()."""
        out = (C.c_double * 3)()
        self._chk(self._lib.sx_device_get_gain_range(self._dev, direction, channel, (name or "").encode(), out))
        return Range(out[0], out[1], out[2])

    def listGains(self, direction, channel):
        return self._str(self._lib.sx_device_list, b"gains", direction).split(",")

    def listAntennas(self, direction, channel):
        return self._str(self._lib.sx_device_list, b"antennas", direction).split(",")

    def readRegisters(self, name, addr, length):
        arr = (C.c_uint * length)()
        self._chk(self._lib.sx_device_read_registers(self._dev, name.encode(), addr, arr, length))
        return list(arr)

    def readRegister(self, name, addr):
        return self.readRegisters(name, addr, 1)[0]

    def writeRegisters(self, name, addr, values):
        arr = (C.c_uint * len(values))(*values)
        self._chk(self._lib.sx_device_write_registers(self._dev, name.encode(), addr, arr, len(values)))

    def writeRegister(self, name, addr, value):
        self.writeRegisters(name, addr, [value])

    def setAntenna(self, direction, channel, name):
        self._chk(self._lib.sx_device_set_antenna(self._dev, direction, channel, name.encode()))

    def getAntenna(self, direction, channel):
        return self._str(self._lib.sx_device_get_antenna, direction, channel)

    # -- streams -----------------------------------------------------------
    def setupStream(self, direction, fmt, channels=(0,), args=None):
        ch = (C.c_size_t * len(channels))(*channels)
        s = self._lib.sx_device_setup_stream(self._dev, direction, fmt.encode(), ch, len(channels), _kwargs(args))
        if not s:
            raise RuntimeError(_err())
        return s

    def closeStream(self, stream):
        self._chk(self._lib.sx_device_close_stream(self._dev, stream))

    def getStreamMTU(self, stream):
        return self._chk(self._lib.sx_device_get_stream_mtu(self._dev, stream))

    def activateStream(self, stream, flags=0, timeNs=0, numElems=0):
        return self._chk(self._lib.sx_device_activate_stream(self._dev, stream, flags, timeNs, numElems))

    def deactivateStream(self, stream, flags=0, timeNs=0):
        return self._chk(self._lib.sx_device_deactivate_stream(self._dev, stream, flags, timeNs))

    @staticmethod
    def _buffs(buffs):
        arr = (C.c_void_p * len(buffs))()
        for i, b in enumerate(buffs):
            if not (isinstance(b, np.ndarray) and b.dtype == np.complex64 and b.flags["C_CONTIGUOUS"]):
                raise TypeError("stream buffers must be contiguous numpy complex64 arrays")
            arr[i] = b.ctypes.data
        return arr

    def readStream(self, stream, buffs, numElems, flags=0, timeoutUs=100000):
        f, t = C.c_int(flags), C.c_longlong(0)
        ret = self._chk(self._lib.sx_device_read_stream(self._dev, stream, self._buffs(buffs), numElems, C.byref(f),
                                                        C.byref(t), timeoutUs))
        return StreamResult(ret, f.value, t.value)

    def writeStream(self, stream, buffs, numElems, flags=0, timeNs=0, timeoutUs=100000):
        f = C.c_int(flags)
        ret = self._chk(self._lib.sx_device_write_stream(self._dev, stream, self._buffs(buffs), numElems, C.byref(f),
                                                         timeNs, timeoutUs))
        return StreamResult(ret, f.value, 0)

    # -- time --------------------------------------------------------------
    def hasHardwareTime(self, what=""):
        return bool(self._chk(self._lib.sx_device_has_hardware_time(self._dev, what.encode())))

    def getHardwareTime(self, what=""):
        t = C.c_longlong(0)
        self._chk(self._lib.sx_device_get_hardware_time(self._dev, what.encode(), C.byref(t)))
        return t.value

    # -- settings / synthetic sink ----------------------------------------------
    def writeSetting(self, key, value):
        self._chk(self._lib.sx_device_write_setting(self._dev, key.encode(), str(value).encode()))

    def readSetting(self, key):
        return self._str(self._lib.sx_device_read_setting, key.encode())

    def txCapture(self, dac_pos, n):
        out = np.empty(n, dtype=np.complex64)
        self._chk(self._lib.sx_device_tx_capture(self._dev, dac_pos, n, out.ctypes.data_as(C.c_void_p)))
        return out
