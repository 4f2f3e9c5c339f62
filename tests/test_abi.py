"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol that include/sxfir.h declares; host-only entry points work; the
product refuses to run without a GPU instead of falling back."""
import os
import re

import numpy as np
import pytest

import sxxcvr_amd
from sxxcvr_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:sxfir|sx_device)_[a-z0-9_]+)\s*\(", text)))


def test_sxfir_exports_every_declared_symbol():
    lib = sxxcvr_amd.load_sxfir()
    names = _declared("sxfir.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libsxfir.so does not export " + n
    # and the python binding declares a prototype for each of them
    assert set(names) <= set(lib._sx_signatures), set(names) - set(lib._sx_signatures)
    assert lib.sxfir_abi_version() == 6


def test_time_arithmetic_matches_oracle(oracle):
    lib = sxxcvr_amd.load_sxfir()
    for clock in (32.0e6, 38.4e6):
        for div in (1536, 768, 512, 256, 128, 64):
            rate = clock / div
            for t in (0, 1, 255, 256, 768, 65536, 10 ** 9 + 7, 123456789012):
                ns = lib.sxfir_ticks_to_time_ns(t, rate)
                assert ns == oracle.ticks_to_time_ns(t, rate)
                assert lib.sxfir_time_ns_to_ticks(ns, rate) == t
                assert lib.sxfir_time_ns_to_ticks(ns + 12345, rate) == oracle.time_ns_to_ticks(ns + 12345, rate)


def test_tap_designer_matches_fixture(golden_dir):
    taps = np.load(os.path.join(golden_dir, "taps.npz"))
    for name, (n, r, g) in {"n128_d4": (128, 4, 1.0), "n256_d8": (256, 8, 1.0), "n256_l8": (256, 8, 8.0),
                            "n1024_d32": (1024, 32, 1.0)}.items():
        got = sxxcvr_amd.design_lowpass(n, r, 8.0, g)
        d = np.abs(got.view(np.int32).astype(np.int64) - taps[name].view(np.int32).astype(np.int64))
        assert d.max() <= 1, name


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, never compute on the host."""
    import ctypes as C
    lib = sxxcvr_amd.load_sxfir()
    n = C.c_int(-1)
    lib.sxfir_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sxxcvr_amd.NativeError) as ei:
        sxxcvr_amd.Resampler(0, np.ones(128, dtype=np.float32), 4)
    assert ei.value.code == -5
    # the gather entry points refuse in the same way (and before librccl is even looked for)
    ident = (C.c_ubyte * 128)()
    assert lib.sxfir_comm_unique_id(ident) == -5
    comm = C.c_void_p()
    assert lib.sxfir_comm_init_rank(C.byref(comm), ident, 1, 0, 0) == -5 and not comm.value
    assert lib.sxfir_comm_gather(None, None, None, 0, 0, 0, 0, None) == -1          # SXFIR_EINVAL: no communicator


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing in the package may reference it."""
    pkg = os.path.join(ROOT, "sxxcvr_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "sx_oracle" not in text and "libsxoracle" not in text and "oracle_lib" not in text, f


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_sxfir", None)
    monkeypatch.setattr(_native, "LIBDIR", str(tmp_path))
    with pytest.raises(ImportError):
        _native.load_sxfir()


def test_header_version_is_the_library_version():
    """One number, written once (include/sxfir.h): the library and the entry point both follow it."""
    import __graft_entry__ as entry
    text = open(os.path.join(ROOT, "include", "sxfir.h")).read()
    want = int(re.search(r"^#define\s+SXFIR_ABI_VERSION\s+(\d+)", text, re.M).group(1))
    assert entry.header_abi_version() == want
    assert sxxcvr_amd.load_sxfir().sxfir_abi_version() == want
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert not re.search(r"sxfir_abi_version\(\)\s*==\s*\d", src), "a literal ABI number in the entry point"


def test_build_entry_point_runs():
    """__graft_entry__.build() end to end in a fresh interpreter: product libraries, the oracle, CMake."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for name in ("libsxfir.so", "libSXSupport.so"):
        assert os.path.exists(os.path.join(ROOT, "sxxcvr_amd", "lib", name))


def test_shipped_code_object():
    """The gfx950 code object inside libsxfir.so (tools/shipped_isa.py: metadata notes + disassembly): no scratch and no
    MFMA anywhere (north star: a short 1-D convolution, not a contraction); the /4 kernel of BASELINE config 2 has its
    packed FMAs with scalar tap operands, non-temporal staging loads for the exclusive rows, no barrier and the resources
    DESIGN.md quotes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import shipped_isa
    rows = shipped_isa.kernels()
    assert len(rows) >= 34
    for r in rows:
        assert r["scratch_bytes"] == 0, r["name"]
        assert r["v_mfma"] == 0, r["name"]
    # an instance holds ONE kind of LDS-DMA: where the typed front end (CF16 storage) writes M0 from inline asm there is no
    # compiler-managed global_load_lds, whose M0 set-up LLVM could hoist or merge across that asm, and no M0 write besides the
    # one in front of each typed instruction (structural in the source: `if constexpr (!HALFIN)`)
    for r in rows:
        if r["typed_lds_dma"]:
            assert r["global_load_lds_dwordx4"] == 0 and r["m0_writes"] == r["typed_lds_dma"], r
    w8 = [r for r in rows if r["name"] == "decim4_wide_kernel<0, false, 24, true, false, 0, false, false>"]
    assert len(w8) == 1, [r["name"] for r in rows]
    w8 = w8[0]
    # the shipped /4 kernel for 128 symmetric taps: 8 outputs per lane x 128 taps = 1024 packed FMAs with scalar taps,
    # a window of 79 chunks read once (+ the output transposition), no barrier, 18 496 B of LDS (8 waves per CU) and a
    # register budget that fits two waves per SIMD; nt loads for all but the two halos
    assert w8["v_pk_fma_f32"] == 1024 and w8["scalar_tap_fmas"] == 1024 and w8["s_barrier"] == 0
    assert w8["vgpr"] <= 256 and w8["lds_bytes"] == 18496 and 79 <= w8["ds_read_b128"] <= 90
    assert 0 < w8["global_load_lds_dwordx4_nt"] < w8["global_load_lds_dwordx4"]
    assert not [r for r in rows if r["name"].startswith("decim4_tile2_kernel")], "round 3's form is a profiling variant now"
    for r in rows:
        if r["name"].startswith("decim_dense_kernel"):
            assert r["lds_bytes"] <= 40960 and r["vgpr"] <= 128 and r["v_pk_fma_f32"] == 512, r    # four workgroups per CU
            # scalar registers spilled into VGPR lanes: a handful around the tile loop at most (the first scalar-tap /8
            # build had 202 of them inside it)
            assert r["sgpr_spill_lane_ops"] <= 40, r
    # CF16 storage at /8, /16, /32 (round 5): the dense kernel's HALFIN instances stage through typed LDS-DMA -- four per 1-KiB instruction of the CF32 form, 36-40 per wave and
    # staging call, two calls in the code (first tile, next tile) -- and convert nothing in the FIR: the only v_cvt_f32_f16 left are
    # the eight of the edge tiles' register path; the multi-column CF16 kernels (184 conversions per 512 FMAs) ship for /4 only
    half = [r for r in rows if r["name"].startswith("decim_dense_kernel<") and r["name"].rstrip(">").endswith("true")]
    assert sorted(r["name"].split("<")[1].split(",")[0] for r in half) == ["16", "32", "8"], [r["name"] for r in half]
    for r in half:
        per_wave = 10 if r["name"].startswith("decim_dense_kernel<32") else 9          # 1-KiB instructions of the CF32 form per wave and tile
        assert r["typed_lds_dma"] == 2 * 4 * per_wave and r["v_cvt_f32_f16"] <= 8 and r["global_load_lds_dwordx4"] == 0, r
    # ... and CF16 at /4 with symmetric taps: the wide kernel with the same front end, 68 typed instructions per tile (one staging site in the tile loop)
    wh = [r for r in rows if r["name"] == "decim4_wide_kernel<0, false, 24, true, false, 0, true, false>"]
    assert len(wh) == 1 and wh[0]["typed_lds_dma"] == 68 and wh[0]["v_cvt_f32_f16"] <= 8 and wh[0]["v_pk_fma_f32"] == 1024, wh
    assert wh[0]["lds_bytes"] == 18496 and wh[0]["scalar_tap_fmas"] == 1024 and wh[0]["global_load_lds_dwordx4"] == 0
    # ... its ASYM form (taps that are not bit-symmetric: P0 taps in VGPR pairs) ships for CF16 storage only, and with it the
    # multi-column kernel has left the production library
    wa = [r for r in rows if r["name"].startswith("decim4_wide_kernel<") and r["name"].rstrip(">").endswith(", true")]
    assert [r["name"] for r in wa] == ["decim4_wide_kernel<0, false, 24, true, false, 0, true, true>"], [r["name"] for r in wa]
    assert wa[0]["vgpr"] <= 256 and wa[0]["typed_lds_dma"] == 68 and wa[0]["v_pk_fma_f32"] == 1024
    assert not [r for r in rows if r["name"].startswith("decim_multi_kernel<")]
    # /48 and /96 (the reference's 50 kS/s and 25 kS/s rates): sixteen-column blocks of whole input lines, two passes of 512 packed
    # FMAs per step, every one with a scalar tap; 70 704 B of LDS (two workgroups per CU), 17 DMA instructions per wave and
    # staging site (two sites), 15 of them non-temporal (the rows no other tile reads); CF32 and wire-word input
    bk = [r for r in rows if r["name"].startswith("decim_blocks_kernel<")]
    # <NB, S32IN, NTLD, HALFIN, SPLIT>: CF32, wire words, CF16 storage (typed LDS-DMA: four per line instruction of the CF32 form);
    # SPLIT (round 6): the instance that deals (tile, block) items, one step per workgroup -- one staging site instead of two, a fifth
    # barrier around the arrival count, half the registers (no tile loop, no waiting block sums), no SGPR spills
    # (a seventh argument, ROT = true: the rotated contract; false exists for a profiling experiment on /16, /32 only)
    # (a sixth argument, RP, round 6: waves by column group, the rows the two windows share kept in registers -- 86 window reads + 4
    # exchange reads where round 5's form, RP = false, had 92 + 8; that form is the profiling build's A/B partner)
    assert sorted(r["name"] for r in bk) == sorted("decim_blocks_kernel<%d, %s, true, %s, %s, true, true>" % (nb, w, hf, sp) for nb in (3, 6)
                                                   for w, hf in (("false", "false"), ("true", "false"), ("false", "true"))
                                                   for sp in ("false", "true")), bk
    for r in bk:
        t = [x.strip() for x in r["name"].split("<")[1].rstrip(">").split(",")]
        half, split = t[3] == "true", t[4] == "true"
        sites = 1 if split else 2
        assert r["lds_bytes"] == 70704 and r["vgpr"] <= 128 and r["v_pk_fma_f32"] == 1024 and r["scalar_tap_fmas"] == 1024, r
        if half:
            # (the typed front end writes M0 from inline asm; such an instance must hold no compiler-managed
            # LDS-DMA, whose M0 set-up LLVM may hoist or merge across the asm)
            # (conversions: the edge tiles' register path only -- a loop of one chunk in the walking form, all 17 chunks unrolled in SPLIT)
            assert r["typed_lds_dma"] == sites * 4 * 17 and r["global_load_lds_dwordx4"] == 0 and r["v_cvt_f32_f16"] <= (4 * 17 if split else 8), r
        else:
            assert r["global_load_lds_dwordx4"] == sites * 17 and r["global_load_lds_dwordx4_nt"] == sites * 15 and r["typed_lds_dma"] == 0, r
        assert r["s_barrier"] == (5 if split else 4) and r["sgpr_spill_lane_ops"] <= 96 and 78 <= r["ds_read_b128"] <= 92, r
        if split:
            assert r["vgpr"] <= 96 and r["sgpr_spill_lane_ops"] <= 4, r      # (an edge item keeps its 17 chunks in flight: 68 registers)
    # interp_tile_kernel ships for CF16 storage only (HALF: typed LDS-DMA front end, half stores) at every ratio of the rate table --
    # x48 / x96 as three phase blocks of its x16 / x32 form; CF32 and wire-word output run the scalar-tap pass kernels
    it = [r for r in rows if r["name"].startswith("interp_tile_kernel<")]
    targs = {r["name"]: [t.strip() for t in r["name"].split("<")[1].rstrip(">").split(",")] for r in it}     # <L, S32OUT, KEYED, LT, HALF>
    assert not [r for r in it if targs[r["name"]][4] == "false"], [r["name"] for r in it]
    assert sorted((int(targs[r["name"]][0]), int(targs[r["name"]][3])) for r in it) == [(4, 4), (8, 8), (16, 16), (16, 48), (32, 32), (32, 96)], it
    for r in it:
        L = int(targs[r["name"]][0])
        tile_in = 16 * (128 // L)                                   # InterpTile<L>::TILE_IN
        assert targs[r["name"]][1:3] == ["false", "false"] and r["typed_lds_dma"] == (tile_in + 32) // 32 and r["global_load_lds_dwordx4"] == 0, r
        nload = ((tile_in + 32) // 2 + 63) // 64                    # the edge tiles' register path: four conversions per chunk and lane
        assert r["v_cvt_f32_f16"] <= 4 * nload and r["v_pk_fma_f32"] == 256, r
    # the scalar-tap pass kernel <QI, KEYED, S32OUT, COUNTED, LL, LT>: x8 (two inputs per lane, four passes), x4 (four inputs, two
    # passes) and x16 .. x96 (two inputs, eight passes per phase block of sixteen; LT / 16 blocks per tile)
    ip = [r for r in rows if r["name"].startswith("interp8_pass_kernel<")]
    pa = {r["name"]: [t.strip() for t in r["name"].split("<")[1].rstrip(">").split(",")] for r in ip}
    # ... and, round 6, a seventh argument PBSPLIT: the x32 / x48 / x96 instances that deal (tile, phase block) items for small calls
    assert sorted((int(pa[n][0]), int(pa[n][4]), int(pa[n][5]), pa[n][6]) for n in pa) == sorted(
        [(2, 8, 8, "false")] * 4 + [(4, 4, 4, "false")] * 4 + [(2, 16, lt, "false") for lt in (16, 32, 48, 96) for _ in range(4)] +
        [(2, 16, lt, "true") for lt in (32, 48, 96) for _ in range(4)]), sorted(pa)
    for r in ip:
        t = pa[r["name"]]
        wire, ll, lt = t[2] == "true", int(t[4]), int(t[5])          # the wire-word conversion holds more masks
        if t[6] == "true":
            lt = ll                                                  # one block per loop iteration: the x16 kernel's counted wait
        assert t[3] == "true", r                                     # the counted form ships
        assert r["lds_bytes"] == {4: 11264, 8: 10240, 16: 18432}[ll] and r["v_pk_fma_f32"] == (512 if ll == 4 else 256), r
        assert r["vgpr"] <= {4: 168, 8: 128, 16: 256}[ll], r         # x16 blocks: LDS (18 KB per wave) holds the CU at 8 waves anyway
        if ll != 16:
            assert r["sgpr_spill_lane_ops"] <= (72 if wire else 16), r
        if lt != ll:
            # phase blocks: the next tile's DMAs are awaited behind the FIRST block's sixteen stores; between the loop's DMAs and that
            # wait the disassembly has stores only (the full-tile path's sixteen and the guarded last-tile path's) and no atomic
            pw = r["phase_block_wait"]
            assert pw["n"] == 16 and pw["waits"] == 1 and pw["atomics_between_loop_dmas_and_wait"] == 0, r
            assert pw["vmem_between_last_dma_and_wait"] == ["global_store_dwordx4"] * 32, r
            continue
        # the counted wait (s_waitcnt vmcnt(8), sxfir_interp_pass.hip.h): safe only if the tile loop issues the next tile's
        # image DMAs, then exactly eight stores and nothing else that counts as VMEM -- no scratch access (checked above), no
        # load, and the keying count's atomic in FRONT of the DMAs.  Read off the shipped disassembly:
        cw = r["counted_wait"]
        n = 16 if ll == 16 else 8                                    # stores per tile = output chunks per lane
        assert cw["n"] == n and cw["waits"] == 1, r
        assert cw["dma_loads"] >= 2 and cw["atomics_after_first_dma"] == 0, r
        assert cw["last_block_vmem"] == ["global_store_dwordx4"] * n, r          # the full-tile path that loops back
        assert cw["non_store_vmem_after_last_dma"] == [], r
    keyed = [r for r in ip if pa[r["name"]][1] == "true"]
    assert len(keyed) == 18


def test_reference_built_checker_is_kept_out_of_history_and_out_of_the_product():
    """oracle/_ref/ (the reference's own converters compiled from /root/reference, `make -C oracle ref`) is git-ignored --
    nothing built from the reference's text enters the history -- and, as the run's contract for it says, NOT gpurun-ignored:
    it travels to the GPU box as a prebuilt checker (tests/test_gpu_kernels.py::
    test_gpu_converters_against_the_reference_compiled_code compares the GPU's converters with it there).  The product never
    touches it: nothing under sxxcvr_amd/, include/ or examples/ names it, and only tests/ load it."""
    ign = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in ign
    gpi = os.path.join(ROOT, ".gpurunignore")
    gpu_ign = open(gpi).read().split() if os.path.exists(gpi) else []
    assert not [l for l in gpu_ign if "_ref" in l], gpu_ign
    import subprocess
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/_ref"], capture_output=True, text=True)
    assert tracked.returncode != 0 or tracked.stdout.strip() == "", tracked.stdout
    hits = []
    for base in ("sxxcvr_amd", "include", "examples"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".inc", ".c")):
                    if "libsxref" in open(os.path.join(d, f), errors="replace").read():
                        hits.append(os.path.join(d, f))
    assert not hits, hits
