#!/bin/bash
# Round 2, GPU call L: scalar-tap kernel with the conflict-free lane mapping.
set -u
OUT=gpurun_out/round2l
mkdir -p $OUT
export TMPDIR=/tmp
true
tail -3 $OUT/pytest.txt
export KB_ROUNDS=15 KB_ITERS=30
timeout 900 python3 tools/kbench.py sb:16:0:0:0 x:16:0:0:0 t2.1.64:16:0:0:0 t2.1.65:16:0:0:0 t2.16.192:1:0:0:0 t2.16.193:1:0:0:0 t2.1.64:32:0:0:0 t2.1.64:8:0:0:0 > $OUT/kbench_ab.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench_ab.txt
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/onekernel.py 4 CF32 28 > /dev/null 2>&1
python3 - "$OUT/pmc" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])) if f else []:
    if "decim" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("per launch (2^28):", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $OUT/pmc
