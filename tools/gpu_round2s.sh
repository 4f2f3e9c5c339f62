#!/bin/bash
# Round 2, GPU call S: which HIP call carries the 6.5 ms outliers of large pageable writeStream calls?
set -u
OUT=gpurun_out/round2s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/devpath_probe.py 1048576 24 > $GRAFT_REPO_ROOT/$OUT/probe.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -4 $OUT/probe.txt | cut -c1-300
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/round2s/trace/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    slow = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])) for r in rows]
    slow.sort(reverse=True)
    print(f, len(rows))
    for d, fn, t in slow[:40]:
        print("%9.1f us %s @%d" % (d / 1e3, fn, t))
for f in glob.glob("gpurun_out/round2s/trace/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys())
    big = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 500000]
    for r in big[:20]:
        print({k: r[k] for k in r if k in ("Direction", "Start_Timestamp", "End_Timestamp", "Bytes", "Size")})
PY
rm -rf $OUT/trace
