#!/bin/bash
# Round 6 A/Bs of the launch geometry at the call sizes the API issues (profiling build, tools/sizebench.py):
#   decim_blocks_kernel (/48, /96): walking form (SXFIR_BLOCKS_SPLIT=0) against (tile, block) items up to n x slots tiles
#   generations of workgroups per launch (SXFIR_OVERSUB) for the dense decimators and the pass interpolators at mid sizes
set -u
export SB_PROF=1
echo "== decim_blocks_kernel: walking form against dealt items while tiles <= n x slots"
for n in 0 2 4 8; do
  echo "-- SXFIR_BLOCKS_SPLIT=$n"
  SXFIR_BLOCKS_SPLIT=$n SB_MODE=rx SB_RATIOS=48,96 SB_LOG2=22,24,25,26,27 python3 tools/sizebench.py CF32 2>&1 | grep "^RX /"
done
echo "== generations per launch (SXFIR_OVERSUB), dense decimators and pass interpolators"
for o in 1 2 4 8 16; do
  echo "-- SXFIR_OVERSUB=$o"
  SXFIR_OVERSUB=$o SB_RATIOS=8,16,32 SB_LOG2=22,23,24,25,26 python3 tools/sizebench.py CF32 2>&1 | grep "^RX /\|^TX x"
done
