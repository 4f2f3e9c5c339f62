#!/bin/bash
# round 6: rocprofv3 kernel trace of the Device path at the reference's slowest rate (which kernels a readStream / writeStream loop
# really launches, and how long they run), then the whole GPU suite once more
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r6j
for r in 25000 75000; do
  DB_RATE=$r rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6j/dev$r -- python3 tools/devbench.py > gpurun_out/r6j/dev$r.log 2>&1
  f=$(find gpurun_out/r6j/dev$r -name "*kernel_stats.csv" | head -1)
  echo "== DB_RATE=$r: $f"; head -12 "$f"
  cp "$f" gpurun_out/r6j/dev${r}_kernel_stats.csv
  find gpurun_out/r6j/dev$r -name "*.csv" -size +4M -delete
done
python3 -m pytest tests -q -m gpu -x > gpurun_out/r6j/tests_all.txt 2>&1
grep -n "passed\|failed" gpurun_out/r6j/tests_all.txt | tail -2
