#!/bin/bash
# Round 2, GPU call H: whole GPU test suite on the new production kernel + bench.py in every configuration.
set -u
OUT=gpurun_out/round2h
mkdir -p $OUT
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
for c in 2 3rx 3tx 5 5h; do
  timeout 300 python3 bench.py --config $c --steps 50 --warmup 20 $( [ $c != 2 ] && echo --no-cpu-baseline --no-through-device ) > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  tail -1 $OUT/bench_$c.json | cut -c1-1500
done
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
tail -1 $OUT/bench_driver_style.json | cut -c1-600
SXFIR_DIST_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --steps 10 --warmup 5 --log2-samples 26 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
tail -1 $OUT/bench_gloo2.json | cut -c1-1200
