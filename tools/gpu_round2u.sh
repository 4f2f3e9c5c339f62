#!/bin/bash
# Round 2, GPU call U: generations (tiles per wave) of the shipped scalar-tap kernel.
set -u
OUT=gpurun_out/round2u
mkdir -p $OUT
export KB_ROUNDS=11 KB_ITERS=30
timeout 900 python3 tools/kbench.py x:16:0:0:0 x:8:0:0:0 x:24:0:0:0 x:32:0:0:0 x:48:0:0:0 x:64:0:0:0 x:16:0:0:2 x:32:0:0:2 x:64:0:0:2 x:16:0:1:0 x:64:0:1:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
