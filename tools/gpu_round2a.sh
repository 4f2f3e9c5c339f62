#!/bin/bash
# Round 2, GPU call A: memory hand-out probe + tile2 variant A/B + correctness of the new variants.
set -u
OUT=gpurun_out/round2a
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o /tmp/membench4 && timeout 300 /tmp/membench4 > $OUT/membench4.txt 2>&1
tail -30 $OUT/membench4.txt
timeout 600 python3 tools/kbench.py sb:16:0:0:0 t2.1.0:16:0:0:0 t2.1.1:16:0:0:0 t2.1.9:16:0:0:0 t2.1.1:64:0:0:2 \
  t2.1.2:64:0:0:2 t2.2.2:64:0:0:2 t2.4.2:64:0:0:2 t2.8.2:64:0:0:2 t2.2.3:16:0:0:0 t2.4.3:16:0:0:0 t2.2.1:16:0:0:0 \
  t2.1.4:16:0:0:0 t2.1.5:16:0:0:0 t2.1.7:16:0:0:0 t2.2.7:16:0:0:0 t2.1.1:32:0:0:0 t2.1.1:8:0:0:0 t2.1.1:4:0:0:0 t2.1.3:16:0:0:0 \
  t2.1.3:32:0:0:0 t2.1.5:4:0:0:0 t2.1.5:64:0:0:0 \
  sb:16:0:1:0 t2.1.0:16:0:1:0 t2.1.1:16:0:1:0 t2.1.2:64:0:1:2 t2.1.5:16:0:1:0 t2.2.3:16:0:1:0 > $OUT/kbench.txt 2>&1
tail -40 $OUT/kbench.txt
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_decim.py tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
