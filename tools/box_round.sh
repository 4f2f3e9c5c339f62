#!/bin/bash
# One fresh box per gpurun call: the headline line as the driver runs it (un-profiled), the same with --settle 0, the
# non-symmetric-taps row, then the rocprofv3 round of config 2 (tools/profile_round.sh).  Everything under gpurun_out/<tag>/.
# Usage: bash tools/box_round.sh <tag>
set -u
TAG=${1:-box}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
rocm-smi --showbus --showuniqueid --showserial > $OUT/box.txt 2>&1
python3 bench.py --no-cpu-baseline --no-through-device --no-rate-table > $OUT/bench_driver_form.json 2> $OUT/bench.err
python3 bench.py --settle 0 --no-cpu-baseline --no-through-device --no-rate-table > $OUT/bench_settle0.json 2>> $OUT/bench.err
python3 bench.py --asymmetric-taps --no-cpu-baseline --no-through-device --no-rate-table > $OUT/bench_asym.json 2>> $OUT/bench.err
bash tools/profile_round.sh $TAG 2 > $OUT/profile_round.log 2>&1
tail -c 300 $OUT/bench_driver_form.json | tr '\n' ' '
