#!/bin/bash
# Round 2, GPU call Y: taps by value in the kernel-argument segment + branch-free lane map (t2.1.576) against the
# shipped kernel (t2.1.64), at 16 / 32 / 64 generations.
set -u
OUT=gpurun_out/round2y
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
export KB_ROUNDS=15 KB_ITERS=30
timeout 900 python3 tools/kbench.py t2.1.64:16:0:0:0 t2.1.576:16:0:0:0 t2.1.64:32:0:0:0 t2.1.576:32:0:0:0 t2.1.64:64:0:0:0 t2.1.576:64:0:0:0 t2.1.576:16:0:0:2 t2.1.576:64:0:0:2 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
