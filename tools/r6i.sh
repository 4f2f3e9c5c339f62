#!/bin/bash
# round 6, after the history carry-over went onto all four waves (decim_blocks_kernel<.., SPLIT>, decim_dense_kernel): the whole GPU suite, the size curve,
# 2^28 against the round-5 tree on the same box
set -u
mkdir -p gpurun_out/r6i
python3 -m pytest tests -q -m gpu -x > gpurun_out/r6i/tests_all.txt 2>&1
tail -2 gpurun_out/r6i/tests_all.txt
python3 tools/sizebench.py CF32 > gpurun_out/r6i/size.txt 2>&1
grep -v amdgpu gpurun_out/r6i/size.txt | grep "^RX" 
{
for rep in 1 2; do
  echo "== round 5 tree, 2^28"; (cd prev_tree && RB_MODE=rx RB_RATIOS=8,16,32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX")
  echo "== this tree, 2^28"; RB_MODE=rx RB_RATIOS=8,16,32,48,96 python3 tools/ratebench.py CF32 2>&1 | grep "RX\|TX"
done
} > gpurun_out/r6i/rates28_ab.txt 2>&1
cat gpurun_out/r6i/rates28_ab.txt
