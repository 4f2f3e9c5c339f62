#!/bin/bash
# What does the board report while the /4 kernel runs back to back?  (power, clocks, temperature, caps)
OUT=gpurun_out/power; mkdir -p $OUT; rm -f $OUT/load.txt
KB_ROUNDS=1500 KB_ITERS=50 python3 tools/kbench.py x:16:0:0:0 > $OUT/kb.txt 2>&1 &
PID=$!
for i in $(seq 1 40); do
  sleep 1
  amd-smi metric -g 0 --power --clock 2>/dev/null | grep -E "SOCKET_POWER|GFX_0:|^ *CLK:|THROTTLE" | head -4 | tr '\n' ' ' >> $OUT/load.txt; echo >> $OUT/load.txt
done
kill $PID 2>/dev/null; wait $PID 2>/dev/null
cat $OUT/load.txt
