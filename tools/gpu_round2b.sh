#!/bin/bash
# Round 2, GPU call B: hand-out sweep of the tile2 kernel, halo carry, priority, phase stamps.
set -u
OUT=gpurun_out/round2b
mkdir -p $OUT
export KB_ROUNDS=7
timeout 900 python3 tools/kbench.py sb:16:0:0:0 \
  t2.1.0:16:0:0:0 t2.1.0:32:0:0:0 t2.1.0:24:0:0:0 t2.1.1:16:0:0:0 t2.1.1:24:0:0:0 t2.1.1:32:0:0:0 t2.1.1:48:0:0:0 \
  t2.1.1:32:0:0:2 t2.1.3:32:0:0:0 t2.2.1:32:0:0:0 t2.2.3:32:0:0:0 t2.2.3:24:0:0:0 t2.2.3:16:0:0:0 t2.2.3:48:0:0:0 t2.2.3:32:0:0:2 \
  t2.1.16:16:0:0:0 t2.1.17:16:0:0:0 t2.1.17:32:0:0:0 t2.1.17:8:0:0:0 t2.1.19:16:0:0:0 t2.2.17:16:0:0:0 t2.2.19:16:0:0:0 t2.2.19:32:0:0:0 \
  t2.1.32:16:0:0:0 t2.1.33:32:0:0:0 t2.2.35:32:0:0:0 t2.1.49:16:0:0:0 t2.2.51:16:0:0:0 \
  t2.1.16:16:0:1:0 t2.1.17:16:0:1:0 t2.1.17:32:0:1:0 t2.1.1:32:0:1:0 t2.2.3:32:0:1:0 > $OUT/kbench_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/kbench_sweep.txt | tail -36
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.0:16:0:5:0 t2.1.1:16:0:5:0 t2.2.3:16:0:5:0 t2.1.17:16:0:5:0 t2.1.5:16:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v amdgpu.ids $OUT/kbench_stamps.txt | tail -12
timeout 600 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
