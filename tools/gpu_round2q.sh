#!/bin/bash
# Round 2, GPU call Q: phase stamps of the multi-column kernel at /32 (CF32, CF16) and /8.
set -u
OUT=gpurun_out/round2q
mkdir -p $OUT
export KB_ROUNDS=5 KB_ITERS=20
for cfg in "32 CF32" "32 CF16" "8 CF32"; do
  set -- $cfg
  echo "== KB_D=$1 KB_FMT=$2" >> $OUT/kbench_multi_stamps.txt
  KB_D=$1 KB_FMT=$2 timeout 300 python3 tools/kbench.py w4:8:0:0:0 w4:8:0:3:0 w4:8:0:1:0 w4:8:0:2:0 w4:2:0:0:0 >> $OUT/kbench_multi_stamps.txt 2>&1
done
grep -v "amdgpu.ids\|checksum" $OUT/kbench_multi_stamps.txt
