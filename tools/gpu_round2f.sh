#!/bin/bash
# Round 2, GPU call F: persistent waves with atomic tile queues.
set -u
OUT=gpurun_out/round2f
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
export KB_ROUNDS=7
timeout 600 python3 tools/kbench.py sb:16:0:0:0 t2.1.0:16:0:0:0 t2.1.64:16:0:0:0 t2.1.65:16:0:0:0 t2.1.128:1:0:0:0 t2.1.129:1:0:0:0 t2.1.192:1:0:0:0 t2.1.193:1:0:0:0 \
   t2.2.128:1:0:0:0 t2.2.193:1:0:0:0 t2.4.129:1:0:0:0 t2.1.128:2:0:0:0 t2.1.193:2:0:0:0 t2.1.128:1:12:0:0 t2.1.193:1:12:0:0 t2.1.128:1:0:1:0 > $OUT/kbench.txt 2>&1
grep -v "amdgpu.ids\|checksum same" $OUT/kbench.txt
for qs in 0 1 3; do SXFIR_QSHIFT=$qs timeout 200 python3 tools/kbench.py t2.1.128:1:0:0:0 t2.1.193:1:0:0:0 2>&1 | grep "^t2" | sed "s/^/qshift=$qs /"; done | tee $OUT/kbench_qshift.txt
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.128:1:0:5:0 t2.1.129:1:0:5:0 t2.1.193:1:0:5:0 t2.1.0:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
