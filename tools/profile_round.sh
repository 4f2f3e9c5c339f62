#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then the two PMC
# passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950) with counters only.
# Usage: bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...
set -u
TAG=${1:-round}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 50 --warmup 50 --no-cpu-baseline"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
tail -1 $OUT/bench.json | cut -c1-400
