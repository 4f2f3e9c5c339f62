#!/bin/bash
# Round 2, GPU call O: the two tap halves on the two waves of a workgroup (decim4_pair_kernel).
set -u
OUT=gpurun_out/round2o
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export KB_ROUNDS=11 KB_ITERS=30
timeout 600 python3 tools/kbench.py sb:16:0:0:0 x:16:0:0:0 pair:16:0:0:0 pair:8:0:0:0 pair:32:0:0:0 pair:4:0:0:0 pair:16:0:0:2 pairx:16:0:0:0 pairx:8:0:0:0 pairx:32:0:0:0 pairx:16:0:0:2 pair:16:0:1:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
export KB_ROUNDS=3 KB_ITERS=10
timeout 300 python3 tools/kbench.py pair:16:0:5:0 pairx:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
