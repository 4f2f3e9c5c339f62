#!/bin/bash
# round 6, one box: parity of the new forms, then same-box before / after against the round-5 tree (prev_tree/, built in the
# container from commit 6c195b9), then the RP A/B of decim_blocks_kernel
set -u
mkdir -p gpurun_out/r6f
(python3 -m pytest tests/test_gpu_variants.py -x -q -k "split or column_group"; python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_decim.py tests/test_gpu_device.py -x -q) > gpurun_out/r6f/tests.txt 2>&1
tail -5 gpurun_out/r6f/tests.txt
{
for rep in 1 2; do
for lg in 20 22 24; do
  echo "== round 5 tree, 2^$lg wideband samples per call"; (cd prev_tree && RB_LOG2N=$lg RB_RATIOS=32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX")
  echo "== this tree, 2^$lg"; RB_LOG2N=$lg RB_RATIOS=32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX"
done
done
} > gpurun_out/r6f/rates_small_ab.txt 2>&1
{
for rep in 1 2; do
for r in 75000 50000 25000; do
  echo "== round 5 tree"; (cd prev_tree && DB_RATE=$r python3 tools/devbench.py 2>&1 | grep "^rate")
  echo "== this tree"; DB_RATE=$r python3 tools/devbench.py 2>&1 | grep "^rate"
done
done
} > gpurun_out/r6f/devbench_ab.txt 2>&1
{
export SB_PROF=1 SB_MODE=rx SB_RATIOS=48,96 SB_LOG2=22,24,28
for rep in 1 2 3; do
for rp in 0 1; do echo "-- SXFIR_BLOCKS_RP=$rp"; SXFIR_BLOCKS_RP=$rp python3 tools/sizebench.py CF32 2>&1 | grep "^RX /" | cut -c1-100; done
done
} > gpurun_out/r6f/rp_ab.txt 2>&1
cat gpurun_out/r6f/rates_small_ab.txt gpurun_out/r6f/devbench_ab.txt gpurun_out/r6f/rp_ab.txt
