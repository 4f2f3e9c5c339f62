#!/bin/bash
# Round 2, GPU call I: where do persistent waves lose time?  Lifetime distribution by XCD / CU / SIMD.
set -u
OUT=gpurun_out/round2i
mkdir -p $OUT
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.64:1:0:5:0 t2.1.64:1:0:5:2 t2.1.64:2:0:5:0 t2.1.64:4:0:5:0 t2.1.64:16:0:5:0 t2.1.0:1:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
