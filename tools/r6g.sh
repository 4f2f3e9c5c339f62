#!/bin/bash
# round 6, second box: the whole GPU suite, the final size curve, same-box before / after at 2^28 against the round-5 tree
# (prev_tree/), the RP A/B again, the Device after the growth cap
set -u
mkdir -p gpurun_out/r6g
python3 -m pytest tests -q -m gpu -x > gpurun_out/r6g/tests_all.txt 2>&1
tail -3 gpurun_out/r6g/tests_all.txt
python3 tools/sizebench.py CF32 > gpurun_out/r6g/size.txt 2>&1
{
for rep in 1 2 3; do
  echo "== round 5 tree, 2^28"; (cd prev_tree && RB_RATIOS=4,8,16,32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX")
  echo "== this tree, 2^28"; RB_RATIOS=4,8,16,32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX"
done
} > gpurun_out/r6g/rates28_ab.txt 2>&1
{
export SB_PROF=1 SB_MODE=rx SB_RATIOS=48,96 SB_LOG2=24,28
for rep in 1 2 3; do
for rp in 0 1; do echo "-- SXFIR_BLOCKS_RP=$rp"; SXFIR_BLOCKS_RP=$rp python3 tools/sizebench.py CF32 2>&1 | grep "^RX /" | cut -c1-100; done
done
} > gpurun_out/r6g/rp_ab.txt 2>&1
{
for r in 75000 50000 25000; do
  echo "== round 5 tree"; (cd prev_tree && DB_RATE=$r python3 tools/devbench.py 2>&1 | grep "^rate")
  echo "== this tree"; DB_RATE=$r python3 tools/devbench.py 2>&1 | grep "^rate"
done
} > gpurun_out/r6g/devbench_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6g/size.txt | tail -22; cat gpurun_out/r6g/rates28_ab.txt gpurun_out/r6g/rp_ab.txt gpurun_out/r6g/devbench_ab.txt
