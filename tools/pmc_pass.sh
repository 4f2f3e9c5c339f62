#!/bin/bash
# usage (on the GPU box): bash tools/pmc_pass.sh "<counters>" <onekernel args...>   -> prints per-launch means
set -u
export TMPDIR=/tmp
C="$1"; shift
OUT=gpurun_out/pmc_$$
timeout 90 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 tools/onekernel.py "$@" > /dev/null 2>&1
python3 - "$OUT" "$*" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])) if f else []:
    if "decim" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $OUT
