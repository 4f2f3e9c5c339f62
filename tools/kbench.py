"""Kernel-variant A/B on one GPU, interleaved rounds in one process (profiling aid; uses the profiling
build of the library, libsxfir_prof.so, whose knobs are environment variables read at plan creation)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sxxcvr_amd
from sxxcvr_amd.resampler import DECIMATE

log2n = int(os.environ.get("KB_LOG2N", "28"))
rounds = int(os.environ.get("KB_ROUNDS", "5"))
iters = int(os.environ.get("KB_ITERS", "10"))
settle = int(os.environ.get("KB_SETTLE", "0"))
nchan = int(os.environ.get("KB_NCHAN", "1"))
D = int(os.environ.get("KB_D", "4"))
FMT = os.environ.get("KB_FMT", "CF32")
configs = []
for spec in (sys.argv[1:] or ["sb:1:0", "db:1:0"]):
    f = spec.split(":")
    v, ov, occ = f[0], f[1], f[2]
    # v:oversub:occ:ablate:sched[:lds_pad]   lds_pad = extra dynamic LDS bytes per workgroup (tile2 variants: fewer waves per CU)
    configs.append((v, int(ov), int(occ), int(f[3]) if len(f) > 3 else 0, int(f[4]) if len(f) > 4 else 0, int(f[5]) if len(f) > 5 else 0))

n = (1 << log2n) // nchan
# CF32 and S32 wire words: 8 bytes per sample in (the decimators always write CF32); CF16: 4 bytes in and out
dt = torch.int32 if FMT == "CF16" else torch.complex64
x = torch.empty((nchan, n), dtype=dt, device="cuda")
sxxcvr_amd.synth_fill(x, 0x51255, 0, 0, fmt=FMT)
if os.environ.get("KB_ZERO") == "1":                 # all-zero input: no toggling in the FMA datapath, the clock stays free of the power cap
    x.zero_()
    print("# all-zero input (KB_ZERO=1): the kernel's structure without the power cap")
if os.environ.get("KB_QBITS"):                       # the same IQ quantised to q fractional bits: fewer toggling mantissa bits per FMA operand
    q = float(1 << int(os.environ["KB_QBITS"]))
    xr = torch.view_as_real(x)
    xr.copy_(torch.round(xr * q) / q)
    print("# input quantised to %s fractional bits (KB_QBITS)" % os.environ["KB_QBITS"])
yoff = int(os.environ.get("KB_YOFF", "0"))          # output buffer displaced by this many bytes (HBM channel phase probe)
ybase = torch.empty(nchan * (n // D) * (4 if FMT == "CF16" else 8) + yoff + 64, dtype=torch.uint8, device="cuda")
y = ybase[yoff:yoff + nchan * (n // D) * (4 if FMT == "CF16" else 8)].view(dt).view(nchan, n // D)
taps = sxxcvr_amd.design_lowpass(32 * D, D)
if os.environ.get("KB_ASYM") == "1":
    # taps that are not bit-symmetric (bench.py --asymmetric-taps): "x" then runs the wide kernel's ASYM form, "sb" the tile kernel
    taps = (taps.astype(np.float64) * (1.0 + 1e-3 * np.arange(taps.size) / taps.size)).astype(np.float32)
plans = []
for v, ov, occ, abl, sched, pad in configs:
    os.environ["SXFIR_LDS_PAD"] = str(pad)
    os.environ["SXFIR_TILE_VARIANT"] = v.replace(".", ":")      # "t2.2.3" -> "t2:2:3" (tile2 kernel: waves per workgroup, option bits)
    os.environ.pop("SXFIR_MULTI_W", None); os.environ.pop("SXFIR_MULTI_PS", None)
    # "densent" / "densepl": the dense kernel with non-temporal / plain staging loads at every ratio; "dense": as shipped
    os.environ.pop("SXFIR_DENSE_NT", None)
    # /8 CF32: "dense" / "x" = as shipped (the scalar-tap subset form), "densev" = the VGPR-tap form; "densent" etc. imply the VGPR form
    os.environ["SXFIR_DENSE_SUBSET"] = "0" if v in ("densev", "densent", "densepl", "densent2") else "1"
    os.environ["SXFIR_DENSE_HC"] = "1" if v == "densehc" else "0"          # densehc: halo carry (/32, /16)
    if v in ("densent", "densepl", "densent2"): os.environ["SXFIR_DENSE_NT"] = {"densent": "1", "densepl": "0", "densent2": "2"}[v]   # densent2: both halos plain
    os.environ["SXFIR_DENSE"] = "0" if v[0] == "w" else "1"   # "dense" (or "x"): decim_dense_kernel at /8, /16, /32; "w4": the multi-column kernel
    if v[0] == "w":                                  # "w4": multi kernel, 4 waves per workgroup; "w8p4": 4-way row split
        w, _, ps = v[1:].partition("p")
        os.environ["SXFIR_MULTI_W"] = w
        if ps: os.environ["SXFIR_MULTI_PS"] = ps
    os.environ["SXFIR_OVERSUB"] = str(ov)
    os.environ["SXFIR_ABLATE"] = str(abl)
    os.environ["SXFIR_SCHED"] = str(sched)
    if occ: os.environ["SXFIR_OCC"] = str(occ)
    else: os.environ.pop("SXFIR_OCC", None)
    plans.append(sxxcvr_amd.Resampler(DECIMATE, taps, D, nchan=nchan, fmt=FMT, profiling=True))
ref = None
res = {c: [] for c in configs}
st = torch.cuda.current_stream().cuda_stream
for r in range(rounds):
    for c, p in zip(configs, plans):
        p.reset()
        try:
            # KB_SETTLE untimed launches of THIS variant first: the chip's power management carries the previous
            # variant's state over for milliseconds, so a short timed run partly measures its predecessor
            if settle: p.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // D, settle, st)
            ms = p.time_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // D, iters, st)
        except Exception as e:                       # e.g. a variant that has no instantiation for this ablation mode
            if r == 0: print("config", c, "skipped:", e)
            res[c].append(float("nan"))
            continue
        res[c].append(ms)
        if r == 0:
            torch.cuda.synchronize()
            chk = (y if FMT == "CF16" else torch.view_as_real(y)).view(torch.int32).sum(dtype=torch.int64).item()
            if ref is None: ref = chk
            print("config", c, "checksum", "same" if chk == ref else "DIFFERENT")
import ctypes as C
for c, p in zip(configs, plans):
    if c[3] in (11, 12):
        mhz = C.c_double()
        if p._lib.sxfir_debug_clock(p._plan, C.byref(mhz)) == 0:
            print("config %s in-kernel shader clock (median over waves): %.0f MHz" % (c, mhz.value))
for c, p in zip(configs, plans):
    if c[3] == 3 and D != 4:
        cap = 1 << 20
        buf = (C.c_ulonglong * (5 * cap))(); nrec = C.c_size_t()
        if p._lib.sxfir_debug_stamps(p._plan, buf, cap, C.byref(nrec)) == 0:
            r = np.frombuffer(buf, dtype=np.uint64, count=5 * nrec.value).reshape(-1, 5).astype(np.float64)
            r = r[r[:, 0] > 0]
            per = r[:, 1:] / r[:, :1]
            print("config %s phases, mean shader cycles per tile per wave over %d waves: reduce+store %.0f | wait data %.0f | "
                  "arithmetic %.0f | barrier + issue next tile's DMAs %.0f | sum %.0f" % (
                      c, len(r), per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean(), per[:, 3].mean(), per.sum(1).mean()))
for c, p in zip(configs, plans):
    if c[3] == 5:                                    # tile2 kernel with phase stamps
        cap = 1 << 20
        buf = (C.c_ulonglong * (8 * cap))(); nrec = C.c_size_t()
        if p._lib.sxfir_debug_stamps(p._plan, buf, cap, C.byref(nrec)) == 0:
            r = np.frombuffer(buf, dtype=np.uint64, count=8 * nrec.value).reshape(-1, 8).astype(np.float64)
            r = r[(r[:, 0] > 0) & (r[:, 6] > 0)]
            per = r[:, 1:5] / r[:, :1]
            mhz = 100.0 * r[:, 5] / r[:, 6]
            print("config %s phases, mean shader cycles per tile per wave over %d waves: issue DMAs %.0f | wait data %.0f | "
                  "FIR %.0f | transpose + stores %.0f | sum %.0f; whole wave %.0f cycles per tile (prologue etc. %.0f per wave); "
                  "tiles per wave %.1f; in-kernel clock p10/p50/p90 %.0f/%.0f/%.0f MHz; wave lifetime %.2f us" % (
                      c, len(r), per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean(), per[:, 3].mean(), per.sum(1).mean(),
                      (r[:, 5] / r[:, 0]).mean(), (r[:, 5] - r[:, 1:5].sum(1)).mean(), r[:, 0].mean(),
                      *np.percentile(mhz, [10, 50, 90]), (r[:, 6] / 100.0).mean()))
            life = r[:, 6] / 100.0
            where = r[:, 7].astype(np.uint64)
            xcc = (where & np.uint64(15)).astype(int)
            hw = (where >> np.uint64(8)).astype(np.uint64)
            simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
            cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
            se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
            print("   wave lifetime us p1/p10/p50/p90/p99/max: %s" % " ".join("%.1f" % v for v in np.percentile(life, [1, 10, 50, 90, 99, 100])))
            print("   mean lifetime by XCC: %s" % " ".join("%d:%.1f" % (k, life[xcc == k].mean()) for k in sorted(set(xcc))))
            cuid = xcc * 1000 + se * 100 + cu
            per_cu = np.array([life[cuid == k].mean() for k in sorted(set(cuid))])
            print("   %d distinct (XCC, SE, CU): mean lifetime per CU p1/p50/p99/max %s; SIMD means %s" % (
                len(per_cu), " ".join("%.1f" % v for v in np.percentile(per_cu, [1, 50, 99, 100])),
                " ".join("%d:%.1f" % (k, life[simd == k].mean()) for k in sorted(set(simd)))))
for c in configs:
    a = np.array(res[c])
    gbs = (8.0 + 8.0 / D) * (0.5 if FMT == "CF16" else 1.0) * (1 << log2n) / (a * 1e-3) / 1e9
    print("%-24s ms med %.4f min %.4f max %.4f | GB/s med %.0f best %.0f | frac of 8TB/s %.3f" % (
        "%s:%d:%d:%d:%d:%d" % c, np.median(a), a.min(), a.max(), np.median(gbs), gbs.max(), np.median(gbs) / 8000))
