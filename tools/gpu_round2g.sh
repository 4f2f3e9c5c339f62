#!/bin/bash
# Round 2, GPU call G: tile queues with the count read at the DMA wait; careful A/B of the candidates.
set -u
OUT=gpurun_out/round2g
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export KB_ROUNDS=5
timeout 300 python3 tools/kbench.py t2.1.128:1:0:0:0 t2.1.193:1:0:0:0 t2.1.129:1:0:0:0 > $OUT/kbench_queue.txt 2>&1
grep "^t2" $OUT/kbench_queue.txt
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.128:1:0:5:0 t2.1.193:1:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep "phases" $OUT/kbench_stamps.txt
export KB_ROUNDS=15 KB_ITERS=30
timeout 900 python3 tools/kbench.py sb:16:0:0:0 t2.1.0:16:0:0:0 t2.1.1:16:0:0:0 t2.1.32:16:0:0:0 t2.1.33:16:0:0:0 t2.2.3:16:0:0:0 t2.2.35:16:0:0:0 \
   t2.1.64:16:0:0:0 t2.1.65:16:0:0:0 t2.1.65:32:0:0:0 t2.1.65:24:0:0:0 t2.2.65:16:0:0:0 t2.2.65:32:0:0:0 t2.1.80:16:0:0:0 t2.1.81:16:0:0:0 > $OUT/kbench_ab.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench_ab.txt
