#!/bin/bash
# Round 2, GPU call C: what clock does the chip hold under the FIR's arithmetic, and where do a wave's cycles go?
set -u
OUT=gpurun_out/round2c
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/valu_power_probe.hip -o /tmp/valu_power_probe 2>/dev/null && timeout 120 /tmp/valu_power_probe > $OUT/valu_power_probe.txt 2>&1
cat $OUT/valu_power_probe.txt
export KB_ROUNDS=3
timeout 300 python3 tools/kbench.py t2.1.0:16:0:5:0 t2.1.1:16:0:5:0 t2.2.3:16:0:5:0 t2.1.17:16:0:5:0 t2.1.5:16:0:5:0 t2.1.0:4:0:5:0 t2.1.0:64:0:5:2 t2.1.0:1:0:5:0 > $OUT/kbench_stamps.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_stamps.txt
