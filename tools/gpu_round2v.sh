#!/bin/bash
# Round 2, GPU call V: decim4_wide_kernel (8 outputs per lane, symmetric SGPR taps, 2 waves/SIMD).
set -u
OUT=gpurun_out/round2v
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export KB_ROUNDS=11 KB_ITERS=30
timeout 900 python3 tools/kbench.py x:16:0:0:0 wide:16:0:0:0 wide2:16:0:0:0 wide4:16:0:0:0 wide12:16:0:0:0 wide16:16:0:0:0 wide:16:0:0:2 wide12:16:0:0:2 wide:32:0:0:0 x:16:0:0:2 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
export KB_ROUNDS=3 KB_ITERS=10
timeout 300 python3 tools/kbench.py wide:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
