#!/bin/bash
# usage (GPU box): bash tools/mb5_pmc.sh <outdir> "<case-name substring>"
# Memory-side counters of the membench5 cases whose name contains the substring; one rocprofv3 pass per counter
# group (the guide: PMC passes on their own, never with a trace), per-launch means per kernel.
set -u
export TMPDIR=/tmp
OUT=$1; F=$2
GROUPS_=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_BUBBLE_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
 "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"
)
i=0
for G in "${GROUPS_[@]}"; do
  D=$OUT/mb5pmc_$i
  timeout 300 rocprofv3 --pmc $G --output-format csv -d $D -- /tmp/membench5 "$F" 6 8 > $D.log 2>&1
  echo "group $i ($G): rc=$?"
  python3 - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])) if f else []:
    k = r["Kernel_Name"]
    k = k[k.find("void ") + 5:] if "void " in k else k
    acc.setdefault((k[:90], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
if not f: print("  no counter file (counter names not accepted on this box?)")
for (k, c), v in sorted(acc.items()):
    print(f"  {k:<92s} {c:<36s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
  rm -rf $D
  i=$((i + 1))
done
