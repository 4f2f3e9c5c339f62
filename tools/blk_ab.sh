#!/bin/bash
set -u
export RB_PROF=1 RB_MODE=rx RB_RATIOS=48,96
for rep in 1 2; do
for o in 1 2 4 8 16; do echo "== oversub $o"; SXFIR_OVERSUB=$o python3 tools/ratebench.py CF32 2>&1 | grep RX; done
echo "== plain loads, oversub 4"; SXFIR_OVERSUB=4 SXFIR_BLOCKS_NT=0 python3 tools/ratebench.py CF32 2>&1 | grep RX
done
