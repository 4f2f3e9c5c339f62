#!/bin/bash
# Round 2, GPU call J: one 16-wave workgroup per CU with an LDS tile queue.
set -u
OUT=gpurun_out/round2j
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export KB_ROUNDS=9 KB_ITERS=20
timeout 600 python3 tools/kbench.py sb:16:0:0:0 t2.1.64:16:0:0:0 t2.16.192:1:0:0:0 t2.16.193:1:0:0:0 t2.16.128:1:0:0:0 t2.16.224:1:0:0:0 t2.8.192:1:0:0:0 t2.4.192:1:0:0:0 \
   t2.16.192:2:0:0:0 t2.16.192:1:0:0:2 t2.16.192:4:0:0:0 t2.16.192:1:0:1:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
export KB_ROUNDS=3 KB_ITERS=10
timeout 300 python3 tools/kbench.py t2.16.192:1:0:5:0 t2.16.128:1:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
