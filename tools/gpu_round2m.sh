#!/bin/bash
# Round 2, GPU call M: whole GPU suite (new chains, whole-stream parity, bench tests) + the bench line.
set -u
OUT=gpurun_out/round2m
mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=12 > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
