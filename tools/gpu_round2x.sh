#!/bin/bash
# Round 2, GPU call X: LDS-DMA lanes of pad slots masked off (t2.1.320) against the shipped kernel (t2.1.64 = x).
set -u
OUT=gpurun_out/round2x
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
export KB_ROUNDS=15 KB_ITERS=30
timeout 900 python3 tools/kbench.py x:16:0:0:0 t2.1.320:16:0:0:0 t2.1.64:16:0:0:0 t2.1.320:64:0:0:0 x:64:0:0:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
