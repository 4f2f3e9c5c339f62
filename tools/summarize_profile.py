#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/<config>/ (tools/profile_round.sh) into the tracked summaries under profiles/:
<name>_<config>_kernel_stats.csv, <name>_<config>_summary.json and the per-config entry of traffic.json.

    python3 tools/summarize_profile.py <tag> <name> [config ...]

PROFILE_SUFFIX=_b appends to the file names (<name>_<config>_b_summary.json: a second box of the same round) and leaves
traffic.json alone.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
configs = sys.argv[3:] or ["2"]
SUFFIX = os.environ.get("PROFILE_SUFFIX", "")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNELS = {"2": "decim4_", "3rx": "decim_dense_kernel<8", "3tx": "interp8_pass_kernel", "5": "decim_dense_kernel<32",
           "5h": "decim_dense_kernel<32, 0, false, 0, false, false, true"}
BYTES = {"2": 10.0, "3rx": 9.0, "3tx": 9.0, "5": 8.25, "5h": 4.125}
# the reference's other rates (bench.py --config rx16 ... tx96)
for _r in (4, 16, 32, 48, 96):
    KERNELS["rx%d" % _r] = "decim_blocks_kernel<%d" % (_r // 16) if _r > 32 else "decim_dense_kernel<%d" % _r
    KERNELS["tx%d" % _r] = "interp8_pass_kernel<2, false, false, true, 16, %d" % _r
    BYTES["rx%d" % _r] = BYTES["tx%d" % _r] = 8.0 + 8.0 / _r
KERNELS["tx4"] = "interp8_pass_kernel<4"


def code_object_registers():
    """VGPR / SGPR / LDS of every kernel as the compiler allocated them (hipcc's resource remarks on the
    production build): rocprofv3's VGPR_Count column is not the code object's figure."""
    cmd = ["hipcc", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-O3", "--cuda-device-only",
           "-c", os.path.join(ROOT, "sxxcvr_amd/csrc/sxfir.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    try:
        err = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
    except Exception:
        return {}
    table, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            dem = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = table.setdefault(re.sub(r"\(.*\)$", "", dem).replace("void ", ""), {})
        elif cur is not None:
            cur[k.split(" ")[0]] = int(v)
    return table


REGS = code_object_registers()
traffic_path = os.path.join(dst, "traffic.json")
try:
    traffic = json.load(open(traffic_path))
    if "configs" not in traffic:
        traffic = {"configs": {}}
except Exception:
    traffic = {"configs": {}}

def duplex_summary(src, name):
    """--config 3: two kernels side by side on two streams; per direction the kernel-trace stats and the counters."""
    def one(pattern):
        f = glob.glob(os.path.join(src, pattern), recursive=True)
        return f[0] if f else None
    out = {"tag": name, "config": "3", "directions": {}}
    stats = one("trace/**/*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats))) if stats else []
    if rows:
        with open(os.path.join(dst, "%s_3_kernel_stats.csv" % name), "w") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
    for d, kern in (("rx", KERNELS["3rx"]), ("tx", KERNELS["3tx"])):
        e = {}
        for r in rows:
            if kern in r["Name"]:
                e.update({"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                          "max_ns": float(r["MaxNs"])})
        pm = {}
        for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_sq"):
            f = one(sub + "/**/*counter_collection.csv")
            if not f:
                continue
            acc = {}
            for r in csv.DictReader(open(f)):
                if kern in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            pm.update({k: sum(v) / len(v) for k, v in acc.items()})
        e["pmc_mean_per_launch"] = pm
        e["algorithmic_bytes_per_launch"] = 9.0 * (1 << 28)
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            fetch, write = 2.0 * pm["FETCH_SIZE"] * 1024.0, pm["WRITE_SIZE"] * 1024.0
            e.update({"hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
                      "traffic_over_algorithmic": (fetch + write) / e["algorithmic_bytes_per_launch"]})
        if "avg_ns" in e:
            e["frac_of_8TBs_at_avg_duration"] = e["algorithmic_bytes_per_launch"] / (e["avg_ns"] * 1e-9) / 8e12
        out["directions"][d] = e
    out["note"] = ("the two kernels overlap in time (two HIP streams), so each one's duration here is its span beside the other; "
                   "counters are collected with the kernels serialised by the profiler")
    b = os.path.join(src, "bench.json")
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            out["bench"] = json.loads(lines[-1])
    json.dump(out, open(os.path.join(dst, "%s_3_summary.json" % name), "w"), indent=1)
    print("3", json.dumps({k: out[k] for k in out if k != "bench"})[:1500])


for cfg in configs:
    src = os.path.join(ROOT, "gpurun_out", tag, cfg)
    if cfg == "3":
        duplex_summary(src, name)
        continue
    kern = KERNELS[cfg]

    def one(pattern):
        f = glob.glob(os.path.join(src, pattern), recursive=True)
        return f[0] if f else None

    out = {"tag": name, "config": cfg}
    stats = one("trace/**/*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(dst, "%s_%s%s_kernel_stats.csv" % (name, cfg, SUFFIX)), "w") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
        for r in rows:
            if kern in r["Name"]:
                out["kernel"] = r["Name"]
                out["calls"] = int(r["Calls"])
                out["avg_ns"] = float(r["AverageNs"])
                out["min_ns"] = float(r["MinNs"])
                out["max_ns"] = float(r["MaxNs"])
    trace = one("trace/**/*kernel_trace.csv")
    if trace:
        rows = [r for r in csv.DictReader(open(trace)) if kern in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            out["trace_row"] = {"VGPR_Count_column": r.get("VGPR_Count"), "SGPR_Count_column": r.get("SGPR_Count"),
                                "lds_bytes": r.get("LDS_Block_Size"), "grid": [r.get("Grid_Size_X"), r.get("Grid_Size_Y")],
                                "workgroup": r.get("Workgroup_Size_X")}
            # the profiled command is bench.py --steps 50 --warmup 50 (default --settle 150).  Its launches of this kernel,
            # in order: 20 after 0.5 s of idle (roofline.kernel_ms_first_20), 150 settle, 50 warm-up steps, the 50 TIMED
            # steps (bracketed by the HIP events bench.py reports as roofline.kernel_ms), 50 from one back-to-back C loop
            # right after the host-side oracle check, about a second of launches beside the telemetry sampler, and 50
            # beside the clock probe.
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            n = len(d)
            if n >= 370:
                mean = lambda a: sum(a) / float(len(a))
                out["avg_ns_by_phase"] = {"first_20_after_idle": mean(d[0:20]), "settle": mean(d[20:170]), "warmup": mean(d[170:220]),
                                          "timed": mean(d[220:270]), "back_to_back_loop": mean(d[270:320]),
                                          "beside_clock_probe": mean(d[n - 50:])}
                if n > 420:
                    out["avg_ns_by_phase"]["while_sampled"] = mean(d[320:n - 50])
                out["launches_by_phase"] = {"first_20_after_idle": 20, "settle": 150, "warmup": 50, "timed": 50, "back_to_back_loop": 50,
                                            "while_sampled": max(0, n - 370), "beside_clock_probe": 50}
    for kname, regs in REGS.items():
        if "kernel" in out and kname.replace("sxfir::", "") in out["kernel"].replace("void ", "").replace("sxfir::", ""):
            out["code_object"] = dict(regs, name=kname)
    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_sq"):
        f = one(sub + "/**/*counter_collection.csv")
        if not f:
            continue
        acc = {}
        for r in csv.DictReader(open(f)):
            if kern not in r.get("Kernel_Name", ""):
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        pm.update({k: sum(v) / len(v) for k, v in acc.items()})
    out["pmc_mean_per_launch"] = pm
    algorithmic = BYTES[cfg] * (1 << 28)
    try:    # the bench line's own figure (ratios 48 and 96 filter a whole number of tiles, a little under 2^28 samples)
        _bl = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")]
        algorithmic = float(json.loads(_bl[-1])["roofline"]["algorithmic_bytes_per_launch"])
    except Exception:
        pass
    out["algorithmic_bytes_per_launch"] = algorithmic
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of
        # the bytes of a wide coalesced streaming read -> double it.
        fetch = 2.0 * pm["FETCH_SIZE"] * 1024.0
        write = pm["WRITE_SIZE"] * 1024.0
        out["hbm_read_bytes_per_launch"] = fetch
        out["hbm_write_bytes_per_launch"] = write
        out["hbm_bytes_per_launch"] = fetch + write
        out["traffic_over_algorithmic"] = (fetch + write) / algorithmic
        traffic["configs"][cfg] = {"hbm_bytes_per_launch": fetch + write, "read": fetch, "write": write,
                                   "source": "profiles/%s_%s_summary.json" % (name, cfg),
                                   "note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction)"}
    if "GRBM_GUI_ACTIVE" in pm and out.get("avg_ns_by_phase"):
        out["grbm_gui_active_over_8_over_duration_mhz"] = pm["GRBM_GUI_ACTIVE"] / 8.0 / out["avg_ns_by_phase"]["timed"] * 1e3
        out["grbm_note"] = ("reads high on 0.5 ms dispatches (MI355X_MICROARCH.md, DVFS): not the shader clock; see "
                            "bench.roofline.gfx_mhz_smi / shader_mhz and profiles/round3_clocks_and_power.txt")
    b = os.path.join(src, "bench.json")
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            out["bench"] = json.loads(lines[-1])
    # The figures a reader needs to recompute the roofline fraction from this file alone: the rocprofv3 durations of this
    # call (all launches / the timed phase) against the algorithmic bytes and the 8 TB/s peak, beside the UN-profiled
    # bench line of the same call on the same box (bench.roofline.frac), and which box that was.
    if "avg_ns" in out:
        out["frac_rocprof_all_launches"] = algorithmic / (out["avg_ns"] * 1e-9) / 8e12
    if out.get("avg_ns_by_phase"):
        out["frac_rocprof_timed"] = algorithmic / (out["avg_ns_by_phase"]["timed"] * 1e-9) / 8e12
    if "bench" in out:
        out["frac_bench_line_same_call"] = out["bench"]["roofline"]["frac"]
        out["box"] = {"gpus": out["bench"]["config"].get("gpus"), "power_cap_w": out["bench"]["roofline"].get("power_cap_w")}
    boxf = os.path.join(src, "box.txt")
    if os.path.exists(boxf):
        out.setdefault("box", {})["rocm_smi"] = [l.strip() for l in open(boxf) if "GPU[" in l][:6]
    json.dump(out, open(os.path.join(dst, "%s_%s%s_summary.json" % (name, cfg, SUFFIX)), "w"), indent=1)
    print(cfg, json.dumps({k: out[k] for k in out if k not in ("bench", "pmc_mean_per_launch")})[:1500])
if not SUFFIX:
    json.dump(traffic, open(traffic_path, "w"), indent=1)
