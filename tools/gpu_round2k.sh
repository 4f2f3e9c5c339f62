#!/bin/bash
# Round 2, GPU call K: final A/B for the production choice of the /4 kernel.
set -u
OUT=gpurun_out/round2k
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export KB_ROUNDS=15 KB_ITERS=30
timeout 900 python3 tools/kbench.py sb:16:0:0:0 t2.1.64:16:0:0:0 t2.1.65:16:0:0:0 t2.16.192:1:0:0:0 t2.16.193:1:0:0:0 t2.16.192:2:0:0:0 t2.16.193:2:0:0:0 t2.16.224:1:0:0:0 t2.16.128:1:0:0:0 t2.16.128:2:0:0:0 > $OUT/kbench_ab.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench_ab.txt
