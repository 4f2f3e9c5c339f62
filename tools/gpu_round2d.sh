#!/bin/bash
# Round 2, GPU call D: scalar-tap kernel (s2) against the tile kernels.
set -u
OUT=gpurun_out/round2d
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "s2 or production" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export KB_ROUNDS=7
timeout 600 python3 tools/kbench.py sb:16:0:0:0 t2.1.0:16:0:0:0 t2.2.3:16:0:0:0 s2.0:16:0:0:0 s2.1:16:0:0:0 s2.0:32:0:0:0 s2.1:32:0:0:0 s2.0:8:0:0:0 s2.0:64:0:0:2 s2.1:24:0:0:0 s2.0:16:0:1:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py s2.0:16:0:5:0 s2.1:16:0:5:0 t2.1.0:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
