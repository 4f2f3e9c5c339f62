#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ profiling directory (tools/profile_round.sh) into the
tracked summaries under profiles/: kernel stats CSV, PMC per-launch figures, traffic.json."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
KERNEL = "decim4_tile_kernel"


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


out = {"tag": name}
stats = one("trace/**/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, name + "_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        if KERNEL in r["Name"]:
            out["kernel"] = r["Name"]
            out["calls"] = int(r["Calls"])
            out["avg_ns"] = float(r["AverageNs"])
            out["min_ns"] = float(r["MinNs"])
            out["max_ns"] = float(r["MaxNs"])
trace = one("trace/**/*kernel_trace.csv")
if trace:
    rows = [r for r in csv.DictReader(open(trace)) if KERNEL in r["Kernel_Name"]]
    if rows:
        r = rows[-1]
        out["vgpr"] = r.get("VGPR_Count")
        out["sgpr"] = r.get("SGPR_Count")
        out["lds_bytes"] = r.get("LDS_Block_Size")
        out["grid"] = [r.get("Grid_Size_X"), r.get("Grid_Size_Y")]
        out["workgroup"] = r.get("Workgroup_Size_X")
        # the profiled command is bench.py --steps 50 --warmup 50: its last 150 launches are warm-up steps,
        # timed steps and the event-timed launches bench.py reports as roofline.kernel_ms; whatever comes
        # before is the untimed settle phase (--settle), which contains the clock transient of the first
        # ~20 launches
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        if len(d) >= 150:
            n = len(d)
            out["avg_ns_by_phase"] = {"warmup": sum(d[n - 150:n - 100]) / 50.0, "timed": sum(d[n - 100:n - 50]) / 50.0,
                                      "event_timed": sum(d[n - 50:]) / 50.0}
            if n > 150:
                out["avg_ns_by_phase"]["settle"] = sum(d[:n - 150]) / float(n - 150)
                out["avg_ns_by_phase"]["settle_first_20"] = sum(d[:20]) / 20.0


def counters(sub):
    f = one(sub + "/**/*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if KERNEL not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    pm.update(counters(sub))
out["pmc_mean_per_launch"] = pm
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # exactly half of the bytes of a wide coalesced streaming read -> double it.
    fetch = 2.0 * pm["FETCH_SIZE"] * 1024.0
    write = pm["WRITE_SIZE"] * 1024.0
    out["hbm_read_bytes_per_launch"] = fetch
    out["hbm_write_bytes_per_launch"] = write
    out["hbm_bytes_per_launch"] = fetch + write
    json.dump({"hbm_bytes_per_launch": fetch + write, "read": fetch, "write": write, "source": name,
               "note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction)"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    lines = [l for l in open(b) if l.startswith("{")]
    if lines:
        out["bench"] = json.loads(lines[-1])
json.dump(out, open(os.path.join(dst, name + "_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
