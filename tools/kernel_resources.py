"""Compact per-kernel resource table (VGPRs, scratch, waves/SIMD, LDS) from hipcc's remarks.  CPU only."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-O3", "--cuda-device-only",
       "-DSXFIR_PROFILING", "-c", os.path.join(ROOT, "sxxcvr_amd/csrc/sxfir.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-5s %-5s %-7s %-4s %-6s kernel" % ("VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    name = re.sub(r"\(.*\)$", "", r["name"]).replace("void sxfir::", "")
    if flt in name:
        print("%-5s %-5s %-7s %-4s %-6s %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS"), name))
