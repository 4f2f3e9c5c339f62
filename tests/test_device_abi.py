"""CPU-side checks of the outer boundary: libSXSupport.so loads, exports every
symbol include/sx_device.h declares, registers "sx" like the reference
(SoapySX.cpp:1629-1656) and refuses to come up without a GPU."""
import os
import re

import pytest

import sxxcvr_amd
import sxxcvr_amd.soapy as SoapySDR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sx_device.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol():
    lib = SoapySDR._load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libSXSupport.so does not export " + n
    assert set(names) <= set(lib._sx_signatures), set(names) - set(lib._sx_signatures)


def test_find_device_is_unconditional():
    # findDevice reports exactly one device whatever the args (SoapySX.cpp:1629-1642)
    found = SoapySDR.Device.enumerate({"driver": "sx"})
    assert found == [{"driver": "sx", "label": "sx"}]
    assert SoapySDR.Device.enumerate("") == found
    assert SoapySDR.Device.enumerate({"driver": "other"}) == []


def test_time_helpers(oracle):
    for rate in (75000.0, 600000.0, 32.0e6 / 768):
        for t in (0, 256, 768, 10 ** 9 + 7):
            assert SoapySDR.ticksToTimeNs(t, rate) == oracle.ticks_to_time_ns(t, rate)
            assert SoapySDR.timeNsToTicks(SoapySDR.ticksToTimeNs(t, rate), rate) == t


def test_make_fails_loudly_without_gpu():
    import ctypes as C
    import sxxcvr_amd
    n = C.c_int(0)
    sxxcvr_amd.load_sxfir().sxfir_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="No MI355X visible"):
        SoapySDR.Device({"driver": "sx"})
    with pytest.raises(RuntimeError, match="no match"):
        SoapySDR.Device({"driver": "nonexistent"})


def test_plugin_is_host_only_cpp():
    """The Device module reaches the GPU only through the extern "C" shim."""
    for f in ("SoapySXHip.cpp", "GpuChains.hpp", "SynthPcm.hpp", "sx_device_capi.cpp"):
        text = open(os.path.join(ROOT, "sxxcvr_amd", "csrc", f)).read()
        assert "hip/hip_runtime" not in text and "hipLaunch" not in text and "<<<" not in text, f


def test_every_device_virtual_is_marked_override():
    """A signature drift against SoapySDR::Device must not compile into a silently non-overriding method: the
    module builds with -Woverloaded-virtual -Wsuggest-override -Werror (against the compat headers here)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not found")
    csrc = os.path.join(ROOT, "sxxcvr_amd", "csrc")
    run = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Woverloaded-virtual", "-Wsuggest-override", "-Werror",
                          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(csrc, "compat"),
                          os.path.join(csrc, "SoapySXHip.cpp")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-3000:]
    text = open(os.path.join(csrc, "SoapySXHip.cpp")).read()
    assert text.count(" override") >= 35


def test_cmake_build_of_the_module(tmp_path):
    """The top-level CMakeLists.txt (find_package(SoapySDR CONFIG) -> SOAPY_SDR_MODULE_UTIL, else the compat
    headers) configures and builds the module against the in-tree libsxfir.so, and the result exports the
    flat C view and the registration entry point."""
    import ctypes
    import shutil
    import subprocess
    cmake = shutil.which("cmake")
    if not cmake:
        pytest.skip("cmake not found")
    libdir = os.path.join(ROOT, "sxxcvr_amd", "lib")
    subprocess.check_call([cmake, "-S", ROOT, "-B", str(tmp_path), "-DSXFIR_PREBUILT_DIR=" + libdir], stdout=subprocess.DEVNULL)
    subprocess.check_call([cmake, "--build", str(tmp_path)], stdout=subprocess.DEVNULL)
    so = os.path.join(str(tmp_path), "libSXSupport.so")
    assert os.path.exists(so)
    sxxcvr_amd.load_sxfir()                              # dependency, RTLD_GLOBAL
    lib = ctypes.CDLL(so)
    for name in ("sx_device_make", "sx_device_read_stream", "sx_device_write_stream"):
        assert hasattr(lib, name), name
