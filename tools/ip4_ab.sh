#!/bin/bash
set -u
# x4: interp8_pass_kernel<4 inputs per lane, L = 4> (shipped) against interp_tile_kernel<4> (SXFIR_IPASS=0, profiling build), generations sweep
export RB_PROF=1 RB_MODE=tx RB_RATIOS=4
for rep in 1 2; do
for o in 2 4 8 16; do
echo "== pass kernel, oversub $o"; SXFIR_OVERSUB=$o python3 tools/ratebench.py CF32 2>&1 | grep TX
echo "== tile kernel, oversub $o"; SXFIR_OVERSUB=$o SXFIR_IPASS=0 python3 tools/ratebench.py CF32 2>&1 | grep TX
done
done
