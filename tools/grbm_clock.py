"""Effective clock from GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel duration, per decimator dispatch:
python3 tools/grbm_clock.py <rocprofv3 output dir> <label>.  MI355X_MICROARCH.md (DVFS): the quotient reads high on
dispatches shorter than about 0.3 ms and comes within 3 % of the in-kernel clock at 10 ms or more."""
import csv, glob, sys
d, label = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc:
    print(label, "no counter file under", d); sys.exit(0)
dur = {}
for r in (csv.DictReader(open(kt[0])) if kt else []):
    if "decim" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for r in csv.DictReader(open(cc[0])):
    if "decim" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        ns = dur.get(r["Dispatch_Id"])
        if ns is None and r.get("End_Timestamp"):
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if ns:
            rows.append((ns, float(r["Counter_Value"]) / 8.0 / ns * 1e3))
for ns, mhz in rows:
    print("log2n=%s dispatch %.3f ms: GRBM_GUI_ACTIVE / 8 / duration = %.0f MHz" % (label, ns / 1e6, mhz))
