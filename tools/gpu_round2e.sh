#!/bin/bash
# Round 2, GPU call E: symmetric scalar-tap variants of the tile2 kernel.
set -u
OUT=gpurun_out/round2e
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export KB_ROUNDS=7
timeout 600 python3 tools/kbench.py sb:16:0:0:0 t2.1.0:16:0:0:0 t2.1.64:16:0:0:0 t2.1.65:16:0:0:0 t2.1.64:32:0:0:0 t2.1.65:32:0:0:0 t2.1.64:8:0:0:0 \
   t2.1.64:64:0:0:2 t2.1.80:16:0:0:0 t2.1.81:16:0:0:0 t2.1.68:16:0:0:0 t2.1.69:16:0:0:0 t2.1.69:4:0:0:0 t2.2.64:16:0:0:0 t2.2.65:16:0:0:0 t2.4.65:16:0:0:0 \
   t2.1.64:16:0:1:0 t2.1.64:16:0:2:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.64:16:0:5:0 t2.1.65:16:0:5:0 t2.1.69:16:0:5:0 t2.1.0:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
