#!/bin/bash
# Runs on the GPU box (via gpurun): for one bench configuration, the bench line, kernel-trace stats of the bench
# command, and the PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950; counters only, never
# together with a trace).
# Usage: bash tools/profile_round.sh <tag> [config ...]     -> gpurun_out/<tag>/<config>/...
set -u
TAG=${1:-round}
shift
CONFIGS=${*:-2}
R=$PWD
export TMPDIR=/tmp
for C in $CONFIGS; do
  OUT=$R/gpurun_out/$TAG/$C
  mkdir -p $OUT
  ARGS="bench.py --config $C --steps 50 --warmup 50 --no-cpu-baseline --no-through-device --no-rate-table"
  rocm-smi --showbus --showuniqueid --showserial > $OUT/box.txt 2>&1
  python3 bench.py --config $C --no-cpu-baseline --no-through-device --no-rate-table > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/pmc_tcc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
  # keep what the summary needs, drop the bulky per-dispatch traces beyond it
  find $OUT -name "*.csv" -size +8M -delete
  tail -1 $OUT/bench.json | cut -c1-300
done
find $R/gpurun_out/$TAG -name "*.csv" | head -40
