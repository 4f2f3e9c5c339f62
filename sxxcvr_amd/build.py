"""In-tree build of the native libraries (hipcc, gfx950 only).

    python -m sxxcvr_amd.build          # build what is out of date
    python -m sxxcvr_amd.build --force

Outputs (git-ignored, shipped to the GPU box by gpurun):
    sxxcvr_amd/lib/libsxfir.so       C ABI of the HIP resampling path (include/sxfir.h)
    sxxcvr_amd/lib/libSXSupport.so   SoapySDR-style Device plugin + its C ABI (include/sx_device.h)
    sxxcvr_amd/lib/sx_devloop, sx_gather_c   plain-C callers of the two ABIs (tools/devloop.c, tools/gather_c.c)
    sxxcvr_amd/lib/libsxfir_prof.so  the same C ABI built with -DSXFIR_PROFILING: kernel A/B variants, ablation
                                     modes and environment knobs (include/sxfir_prof.h); tools/ and
                                     tests/test_gpu_variants.py only, never loaded by the product
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the product has no non-HIP build")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(*dirs):
    out = []
    for d in dirs:
        for base, _, files in os.walk(d):
            out += [os.path.join(base, f) for f in files if f.endswith((".hip", ".h", ".hpp", ".cpp", ".inc"))]
    return out


# libsxfir.so: HIP kernels + the extern "C" shim, built by hipcc for gfx950.
# libSXSupport.so: host-only C++ (no HIP headers): the SoapySDR Device module, its flat C view and,
# because the real SoapySDR headers/library are absent in this image, the API-compatible subset under
# csrc/compat.  It reaches the GPU only through libsxfir.so's C ABI.
TARGETS = {
    "libsxfir.so": {
        "compiler": "hipcc",
        "sources": ["sxfir.hip"],
        "flags": ["--offload-arch=" + ARCH, "-O3"],
        "libs": ["-ldl"],           # librccl is dlopen'ed by sxfir_comm_* on first use, never linked
    },
    "libsxfir_prof.so": {
        "compiler": "hipcc",
        "sources": ["sxfir.hip"],
        "flags": ["--offload-arch=" + ARCH, "-O3", "-DSXFIR_PROFILING"],
        "libs": ["-ldl"],
    },
    "libSXSupport.so": {
        "compiler": "g++",
        "sources": ["SoapySXHip.cpp", "sx_device_capi.cpp", "compat/SoapySDRCompat.cpp"],
        "flags": ["-O2", "-ffp-contract=off", "-pthread", "-I" + os.path.join(CSRC, "compat")],
        "libs": ["-L" + LIBDIR, "-lsxfir", "-Wl,-rpath,$ORIGIN"],
    },
    # a C caller of the flat Device API (include/sx_device.h): per-call cost of readStream / writeStream with no
    # Python in the loop; bench.py runs it as a child process (through_device.c_caller)
    "sx_devloop": {
        "compiler": "gcc",
        "executable": True,
        "sources": [os.path.join(ROOT, "tools", "devloop.c")],
        "flags": ["-O2"],
        "libs": ["-L" + LIBDIR, "-lSXSupport", "-Wl,-rpath,$ORIGIN"],
    },
    # a C caller of the inner ABI (include/sxfir.h): the sharded path with the RCCL gather driven from plain C
    # (sxfir_comm_*), rank-per-process or one process for all GPUs; tests/test_gpu_rccl.py and bench.py --gather capi
    "sx_gather_c": {
        "compiler": "gcc",
        "executable": True,
        "sources": [os.path.join(ROOT, "tools", "gather_c.c")],
        "flags": ["-O2"],
        "libs": ["-L" + LIBDIR, "-lsxfir", "-Wl,-rpath,$ORIGIN"],
    },
}


def git_commit():
    """HEAD as the reference's version.sh prints it (SoapySX/version.sh: `git rev-parse HEAD`, "-dirty" when the index
    differs): what getHardwareInfo()["soapysx_commit"] and makeDevice's log line report (SoapySX.cpp:1577-1581, :1649).
    None where there is no git checkout (the GPU box's snapshot): the library built in the container is kept."""
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(["git", "-C", ROOT, "diff-index", "--quiet", "HEAD", "--"], stderr=subprocess.DEVNULL) != 0
        return head + ("-dirty" if dirty else "")
    except (OSError, subprocess.CalledProcessError):
        return None


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    common = _deps(CSRC, os.path.join(ROOT, "include"))
    built = []
    commit = git_commit()
    stamp_file = os.path.join(LIBDIR, ".soapysx_commit")
    stamped = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
    for name, spec in TARGETS.items():
        srcs = [s if os.path.isabs(s) else os.path.join(CSRC, s) for s in spec["sources"]]
        if not all(os.path.exists(s) for s in srcs):
            continue
        out = os.path.join(LIBDIR, name)
        # sources outside csrc/ (the C caller under tools/) are dependencies of their own target only
        deps = common + [s for s in srcs if not s.startswith(CSRC + os.sep)]
        restamp = name == "libSXSupport.so" and commit is not None and commit != stamped
        if not (force or restamp or _newer(out, deps)):
            continue
        if spec.get("executable"):
            cmd = [shutil.which("gcc") or "gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-Wall",
                   "-I" + os.path.join(ROOT, "include")] + spec["flags"] + srcs + ["-o", out] + spec.get("libs", [])
        else:
            cc = hipcc() if spec["compiler"] == "hipcc" else (shutil.which("g++") or "g++")
            stamp = ['-DSOAPYSX_COMMIT="%s"' % (commit or "unknown")] if name == "libSXSupport.so" else []
            cmd = [cc, "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                   "-I" + os.path.join(ROOT, "include")] + spec["flags"] + stamp + srcs + ["-o", out] + spec.get("libs", [])
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(out)
        if name == "libSXSupport.so":
            with open(stamp_file, "w") as f:
                f.write(commit or "unknown")
    return built


if __name__ == "__main__":
    b = build(force="--force" in sys.argv, verbose=True)
    print("built:", b if b else "(up to date)")
