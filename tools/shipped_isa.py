"""What actually ships: the gfx950 code object inside sxxcvr_amd/lib/libsxfir.so, per kernel (CPU only).

    python3 tools/shipped_isa.py [substring of the kernel name] [--json]

Registers / LDS / scratch come from the code object's metadata notes, instruction counts from its disassembly:
v_pk_fma_f32, ds_read_b128, global_load_lds_dwordx4 (and how many of those carry `nt`), s_barrier, v_mfma, DPP and
permlane ops.  For the /4 scalar-tap kernel also the issue order of its packed FMAs: the share of adjacent FMAs
that use the same sample pair (source operand 1) -- 0.75 when every four FMAs of a sample pair are back to back
(T2_XGROUP as written), 0.0 when the two sample streams alternate.  DESIGN.md quotes this table;
tests/test_abi.py::test_shipped_code_object checks its invariants (no scratch, no MFMA, nt loads present, FMA
grouping not lost to a compiler update).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def extract(lib):
    d = tempfile.mkdtemp(prefix="sxisa_")
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def kernels(lib=None):
    lib = lib or os.path.join(ROOT, "sxxcvr_amd", "lib", "libsxfir.so")
    co = extract(lib)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    meta = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        meta[name.group(1)] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "lds_bytes": g("group_segment_fixed_size"),
                               "scratch_bytes": g("private_segment_fixed_size")}
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
    pretty = demangle(list(meta))
    rows = []
    for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^\n|\Z)", asm, re.S | re.M):
        sym, body = m.group(1), m.group(2)
        if sym not in meta:
            continue
        ops = re.findall(r"^\s+(\S+)", body, re.M)
        cnt = lambda pat: sum(1 for o in ops if re.match(pat, o))
        r = dict(meta[sym])
        r["name"] = re.sub(r"\(.*\)$", "", pretty[sym]).replace("void sxfir::", "")
        r.update({"v_pk_fma_f32": cnt(r"v_pk_fma_f32$"), "ds_read_b128": cnt(r"ds_read_b128$"),
                  "global_load_lds_dwordx4": cnt(r"global_load_lds_dwordx4$"),
                  "global_load_lds_dwordx4_nt": len(re.findall(r"global_load_lds_dwordx4[^\n]*\bnt\b", body)),
                  "s_barrier": cnt(r"s_barrier$"), "v_mfma": cnt(r"v_mfma"), "dpp_or_permlane": len(re.findall(r"_dpp|v_permlane", body)),
                  "s_load_dwordx16": cnt(r"s_load_dwordx16$"), "instructions": len(ops),
                  "valu": cnt(r"v_"), "sgpr_spill_lane_ops": cnt(r"v_(read|write)lane_b32$"),
                  # CF16 storage: typed LDS-DMA (the texture path converts on the way in) against conversions by the VALU
                  "typed_lds_dma": len(re.findall(r"buffer_load_format_x [^\n]*\blds\b", body)), "v_cvt_f32_f16": cnt(r"v_cvt_f32_f16"),
                  # writes of M0 (the LDS address of an LDS-DMA): one per typed instruction where the front end writes it from
                  # inline asm -- and then none of the compiler's own (an instance holds one kind of LDS-DMA or the other)
                  "m0_writes": len(re.findall(r"^\s*s_mov_b32 m0,", body, re.M))})
        fm = re.findall(r"v_pk_fma_f32 v\[\d+:\d+\], (s\[\d+:\d+\]), (v\[\d+:\d+\])", body)
        if len(fm) > 1:
            r["scalar_tap_fmas"] = len(fm)
            r["adjacent_fmas_sharing_sample_pair"] = round(sum(1 for a, b in zip(fm, fm[1:]) if a[1] == b[1]) / (len(fm) - 1), 3)
        # (the compiler emits vmcnt(N > 0) waits of its own elsewhere: those count its own loads; this one is hand-written)
        if "interp8_pass_kernel" in r["name"]:
            targs = [t.strip() for t in r["name"].split("<")[1].rstrip(">").split(",")]      # <QI, KEYED, S32OUT, COUNTED, LL, LT, PBSPLIT>
            if targs[4] == targs[5] or targs[6] == "true":           # one block per loop iteration: the wait at the loop head
                cw = counted_wait(body)
                if cw:
                    r["counted_wait"] = cw
            else:
                r["phase_block_wait"] = phase_block_wait(body)
        rows.append(r)
    return rows


def counted_wait(body):
    """A kernel that waits with `s_waitcnt vmcnt(N)`, N > 0 (interp8_pass_kernel: the next tile's image DMAs are issued in
    front of this tile's N stores, and "at most N outstanding" then means the DMAs have landed) relies on the exact VMEM
    instruction sequence of its tile loop.  Returns what the disassembly says about it, in code order from that wait to
    the end of the kernel (the loop body: window reads, [keying atomic], staging of the next tile, FMAs, stores, branch
    back): the VMEM mnemonics, and the facts the wait presumes."""
    lines = [l.split("//")[0].strip() for l in body.splitlines()]
    lines = [l for l in lines if l]
    w = [i for i, l in enumerate(lines) if re.match(r"s_waitcnt vmcnt\(([1-9]\d*)\)", l)]
    if not w:
        return None
    n = int(re.match(r"s_waitcnt vmcnt\((\d+)\)", lines[w[0]]).group(1))
    tail = lines[w[0]:]
    vm = [(i, l.split()[0]) for i, l in enumerate(tail) if re.match(r"(global|buffer|scratch|flat)_", l)]
    dma = [i for i, o in vm if o.startswith("global_load_lds")]
    atom = [i for i, o in vm if "atomic" in o]
    # the last basic block: after the last conditional branch, up to the branch back to the loop head
    back = max(i for i, l in enumerate(tail) if l.startswith("s_branch"))
    cond = max(i for i, l in enumerate(tail[:back]) if l.startswith("s_cbranch"))
    last_block = [o for i, o in vm if cond < i < back]
    after_dma = [o for i, o in vm if dma and i > max(dma)]
    return {"n": n, "waits": len(w), "dma_loads": len(dma),
            "atomics_after_first_dma": sum(1 for i in atom if dma and i > min(dma)),
            "last_block_vmem": last_block,
            "non_store_vmem_after_last_dma": [o for o in after_dma if not o.startswith("global_store")]}


def phase_block_wait(body):
    """interp8_pass_kernel over phase blocks (x32, x48, x96): the next tile's image DMAs are awaited behind the FIRST block's
    stores with `s_waitcnt vmcnt(16)`.  In code order the tile loop is: window reads, [keying atomic], the DMAs, then the block
    loop -- passes, the block's stores (a full-tile path of sixteen and a guarded last-tile path), the wait.  Returns the VMEM
    mnemonics between the loop's last DMA and the hand-written wait -- they must all be stores, sixteen per path -- and the
    atomics between the loop's DMAs and the wait (none: the keying count's atomic sits in front of the DMAs)."""
    lines = [l.split("//")[0].strip() for l in body.splitlines()]
    lines = [l for l in lines if l]
    w = [i for i, l in enumerate(lines) if re.match(r"s_waitcnt vmcnt\(16\)$", l)]
    dma = [i for i, l in enumerate(lines) if l.startswith("global_load_lds")]
    if not w or not dma:
        return None
    after = [i for i in w if i > max(dma)]
    if not after:
        return None
    between = [l.split()[0] for l in lines[max(dma) + 1:after[0]] if re.match(r"(global|buffer|scratch|flat)_", l)]
    atom = [i for i, l in enumerate(lines) if l.startswith(("global_", "flat_", "buffer_")) and "atomic" in l.split()[0]]
    # the tile loop's own DMAs are the last group in code order (the first group stages the workgroup's first tile)
    loop_dma = [i for i in dma if i > max(atom)] if atom else dma[len(dma) // 2:]
    return {"n": 16, "waits": len(w), "dma_loads": len(dma), "vmem_between_last_dma_and_wait": between,
            "atomics_between_loop_dmas_and_wait": sum(1 for i in atom if loop_dma and min(loop_dma) < i < after[0])}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = args[0] if args else ""
    rows = [r for r in kernels() if flt in r["name"]]
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
    else:
        print("%-5s %-5s %-6s %-7s %-6s %-5s %-8s %-5s %-5s %-6s %-6s %-5s %-5s kernel" % ("VGPR", "SGPR", "LDS", "scratch", "pkfma", "ds128", "dma(nt)", "barr", "mfma", "xshare", "instr", "valu", "spill"))
        for r in sorted(rows, key=lambda r: r["name"]):
            print("%-5d %-5d %-6d %-7d %-6d %-5d %-8s %-5d %-5d %-6s %-6d %-5d %-5d %s" % (
                r["vgpr"], r["sgpr"], r["lds_bytes"], r["scratch_bytes"], r["v_pk_fma_f32"], r["ds_read_b128"],
                "%d(%d)" % (r["global_load_lds_dwordx4"], r["global_load_lds_dwordx4_nt"]), r["s_barrier"], r["v_mfma"],
                r.get("adjacent_fmas_sharing_sample_pair", "-"), r["instructions"], r["valu"], r["sgpr_spill_lane_ops"], r["name"]))
