#!/bin/bash
# One script for the GPU calls of a round (replaces the per-call run logs of round 2):
#     gpurun --timeout 900 -- 'bash tools/gpu_steps.sh <tag> <step> [<step> ...]'
# Each step appends to gpurun_out/<tag>/<step>.txt; what is worth keeping is copied to profiles/ by hand.
# Steps (arguments after ':' are passed on, ',' separated):
#   tests[:expr]        pytest -m gpu (optionally -k expr)
#   testsall            the whole GPU suite without -x;  repeat:expr  the selected tests five times (flakiness)
#   bench[:config]      bench.py --config C (default 2) without the CPU legs
#   benchfull           bench.py with everything (the driver's call)
#   kb32 | kb32h | kb8 | kb16   kbench A/B of the /32 (CF32, CF16), /8, /16 kernels incl. ablations and stamps
#   kb4                 the /4 production kernel + its memory side;  kb4o  FMA issue-order variants of it
#   ib8                 interpolator x8 (tiled)
#   pmc:D[,fmt]         LDS / VALU counters of one decimator shape (tools/pmc_pass.sh)
#   profile:config      tools/profile_round.sh for one bench configuration
#   clocks              tools/clock_calib.hip: shader clock by instruction count vs s_memtime / s_memrealtime
#   pprobe[:D]          tools/power_probe.py: board power, caps and clocks, random vs all-zero input
#   grbm                GRBM_GUI_ACTIVE-derived clock on a 0.5 ms and a 33 ms dispatch of the /4 kernel
#   psplit[:D]          tools/power_split.py: power of the whole kernel / memory side alone / arithmetic alone
#   mempower            tools/mempower.py: power beside plain streaming kernels (4:1 via registers / LDS-DMA, read, copy)
#   kb4g kb4p kb4t kb4c8  /4 kernel: generations sweep, pipelined LDS reads, tap-major FMA order, 8-channel layout
#   soak[:n]            the whole GPU suite n times (default 16), outcome lines counted
#   devpath:n,..        tools/devpath_probe.py per-call times at the given block sizes
#   power               board power / clock while the /4 kernel runs (tools/gpu_power.sh)
#   kb4nt               /4 kernel: non-temporal staging loads (all DMAs / all but the next tile's halo) against the shipped kernel, and their memory sides
#   hostvis[:iters[,filter]]  tools/hostvis_probe.hip: kernel stores into host memory, every word checked after the wait
#   gatherc[:log2n]     tools/gather_c.c: the C-ABI gather (sxfir_comm_*) on one rank, both forms
#   kb4w                /4 kernel: 8 against 16 waves per CU (LDS padding), nt loads, whole kernel and memory side, random and all-zero input
#   valu                tools/valu_power_probe.hip: the FIR's arithmetic alone by operand source and LDS read count
#   kb4x                /4 kernel: tiles per wave x nt loads x taps by value x deferred stores x 12 waves per CU x pinned FMA order
#   kbdnt               dense /32, /8, /16 kernels with non-temporal staging loads against the shipped ones (whole kernel, memory side)
#   kb4q                /4 kernel: persistent workgroups with an LDS tile queue + nt loads against generations of short waves
#   kb4f                /4 kernel: generations / dispatch order / nt mask around the shipped configuration
#   kb4wide             /4 kernel: eight outputs per lane + nt loads against the shipped form
#   ib8p                x8 interpolator: interp8_pass_kernel (scalar taps) against interp_tile_kernel, long visits
#   mb5[:filter]        tools/membench5.hip: the round-4 streaming sweep (shape x bytes in flight x cache policy)
#   mb5pmc:filter       memory-side counters (TCC_EA0_*, TCP_PENDING_STALL, SQ_WAIT_INST_ANY ...) of the cases matching filter
#   listpmc             rocprofv3 --list-avail (which counters this box exposes)
#   kbq                 the shipped /4 and /32 kernels on IQ quantised to 1..16 fractional bits: time against operand entropy
#   kbhc                dense /32, /16 with halo carry (runs of consecutive tiles, halo copied inside LDS) against the shipped form
#   walls               every shipped kernel on random IQ, on zeros and its memory side alone (long visits, one box)
#   prices[:seconds]    tools/price_list.py: dynamic energy per launch of the arithmetic probe's mixes -> nJ per extra instruction
#   psplit4w            socket power of the shipped /4 kernel, its memory side alone, round 3's form, and plain streams
#   kb4pol              /4 shipped kernel against the same build with other cache policies (sc0 / sc1 / nt) on loads and stores
#   ibprev              x8 interpolator: the same before / after for interp8_pass_kernel
#   kbprev:D[,variant[,fmt]]  the previous commit's profiling library (tools/prev_lib.sh) against this tree's, alternating processes
# Every step's exit status is recorded (rc=N in its log and on stdout); the script exits non-zero if any step
# failed or was unknown, so `gpurun -- 'bash tools/gpu_steps.sh ...'` reports a failing GPU suite.
set -u -o pipefail
FAILED=""
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp KB_ROUNDS=${KB_ROUNDS:-5} KB_ITERS=${KB_ITERS:-20}
for S in "$@"; do
  NAME=${S%%:*}; ARG=""; [ "$S" != "$NAME" ] && ARG=${S#*:}
  LOG=$OUT/$(echo $S | tr ':,/ ' '____').txt
  echo "=== $S" | tee -a $LOG
  RC=0
  case $NAME in
    tests)    if [ -n "$ARG" ]; then timeout 1500 python3 -m pytest tests -m gpu -x -q -k "$ARG" >> $LOG 2>&1; else timeout 2400 python3 -m pytest tests -m gpu -x -q >> $LOG 2>&1; fi; RC=$?; tail -5 $LOG ;;
    testsall) timeout 2700 python3 -m pytest tests -m gpu -q >> $LOG 2>&1; RC=$?; tail -15 $LOG ;; # no -x: every failure
    repeat)   for i in 1 2 3 4 5; do timeout 600 python3 -m pytest tests -m gpu -q -k "$ARG" 2>&1 | tail -3 >> $LOG; done; RC=$?; cat $LOG ;;
    soak)     for i in $(seq 1 ${ARG:-16}); do timeout 900 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|differ|Error" | cut -c1-300 >> $LOG; done; RC=$?; sort $LOG | uniq -c | tail -8 ;;
    repfile)  for i in 1 2 3 4 5 6 7 8; do timeout 900 python3 -m pytest $ARG -m gpu -q 2>&1 | grep -E "passed|failed|differ" | cut -c1-400 >> $LOG; done; RC=$?; cat $LOG ;;
    devpath)  for n in ${ARG//,/ }; do timeout 300 python3 tools/devpath_probe.py $n 60 >> $LOG 2>&1; done; RC=$?; grep -v amdgpu.ids $LOG | tail -16 ;;
    stress)   timeout 900 python3 tools/stress_direct_rx.py ${ARG//,/ } >> $LOG 2>&1; RC=$?; grep -v amdgpu.ids $LOG | tail -25 ;;
    bench)    timeout 600 python3 bench.py --config ${ARG:-2} --no-cpu-baseline --no-through-device >> $LOG 2>&1; RC=$?; tail -1 $LOG | cut -c1-1500 ;;
    benchfull) T0=$(date +%s); timeout 900 python3 bench.py >> $LOG 2>&1; RC=$?; echo "bench.py wall seconds: $(( $(date +%s) - T0 ))" | tee -a $LOG; tail -2 $LOG | cut -c1-3000 ;;
    kb32)     KB_D=32 timeout 600 python3 tools/kbench.py dense:8:0:0:0 w4:8:0:0:0 dense:8:0:3:0 w4:8:0:3:0 dense:8:0:1:0 dense:8:0:2:0 w4:8:0:1:0 w4:8:0:2:0 dense:4:0:0:0 dense:16:0:0:0 dense:2:0:0:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -40 ;;
    kb32h)    KB_D=32 KB_FMT=CF16 timeout 600 python3 tools/kbench.py w4:8:0:0:0 w4:8:0:3:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -12 ;;
    kb8)      KB_D=8 timeout 600 python3 tools/kbench.py dense:8:0:0:0 w4:8:0:0:0 dense:8:0:3:0 w4:8:0:3:0 dense:8:0:1:0 dense:8:0:2:0 w4:8:0:1:0 w4:8:0:2:0 dense:4:0:0:0 dense:16:0:0:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -24 ;;
    kb16)     KB_D=16 timeout 600 python3 tools/kbench.py dense:8:0:0:0 w4:8:0:0:0 dense:8:0:3:0 w4:8:0:3:0 dense:8:0:1:0 dense:8:0:2:0 w4:8:0:1:0 w4:8:0:2:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -20 ;;
    kb4)      KB_D=4 timeout 600 python3 tools/kbench.py x:16:0:0:0 t2.1.1088:16:0:5:0 t2.1.1088:16:0:1:0 t2.1.1088:16:0:2:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -14 ;;
    kb4g)     KB_D=4 KB_ROUNDS=7 timeout 900 python3 tools/kbench.py x:16:0:0:0 x:8:0:0:0 x:32:0:0:0 x:64:0:0:0 x:24:0:0:0 t2.1.1088:16:0:1:0 t2.1.1088:64:0:1:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -9 ;;
    kb4p)     KB_D=4 KB_ROUNDS=9 timeout 900 python3 tools/kbench.py t2.1.1088:16:0:0:0 t2.1.5184:16:0:0:0 t2.1.13376:16:0:0:0 t2.1.9280:16:0:0:0 t2.1.1088:16:0:5:0 t2.1.9280:16:0:5:0 t2.1.5184:16:0:5:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep -v "wave lifetime\|mean lifetime\|distinct" | tail -14 ;;
    kb4c8)    KB_D=4 KB_NCHAN=8 timeout 600 python3 tools/kbench.py x:16:0:0:0 x:8:0:0:0 x:32:0:0:0 >> $LOG 2>&1; KB_D=4 KB_NCHAN=1 timeout 600 python3 tools/kbench.py x:16:0:0:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -6 ;;
    kb4t)     KB_D=4 KB_ROUNDS=9 timeout 900 python3 tools/kbench.py t2.1.1088:16:0:0:0 t2.1.16448:16:0:0:0 t2.1.64:16:0:0:0 x:16:0:0:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -9 ;;
    kb4o)     KB_D=4 KB_ROUNDS=9 timeout 900 python3 tools/kbench.py t2.1.64:16:0:0:0 t2.1.1088:16:0:0:0 t2.1.2112:16:0:0:0 x:16:0:0:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -10 ;;
    ib8)      KB_L=8 timeout 600 python3 tools/ibench.py >> $LOG 2>&1; RC=$?; tail -8 $LOG ;;
    pmc)      bash tools/pmc_pass.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" ${ARG//,/ } >> $LOG 2>&1; RC=$?; tail -2 $LOG ;;
    profile)  bash tools/profile_round.sh $TAG ${ARG:-2} >> $LOG 2>&1; RC=$?; tail -3 $LOG ;;
    clocks)   hipcc --offload-arch=gfx950 -O3 tools/clock_calib.hip -o /tmp/clock_calib >> $LOG 2>&1 && timeout 120 /tmp/clock_calib >> $LOG 2>&1; RC=$?; tail -9 $LOG ;;
    mempower) timeout 300 python3 tools/mempower.py >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -5 ;;
    psplit)   timeout 300 python3 tools/power_split.py ${ARG:-4} >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -4 ;;
    pprobe)   timeout 300 python3 tools/power_probe.py ${ARG:-4} >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -6 ;;
    grbm)     # GRBM_GUI_ACTIVE on a short (2^28) and a long (2^34 samples, ~33 ms) dispatch of the same kernel
              for L in 28 34; do timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm_$L -- python3 tools/onekernel.py 4 CF32 $L > /dev/null 2>&1; python3 tools/grbm_clock.py $OUT/grbm_$L $L >> $LOG 2>&1; done; find $OUT -name "*.csv" -size +1M -delete; tail -4 $LOG ;;
    power)    bash tools/gpu_power.sh >> $LOG 2>&1; RC=$?; tail -20 $LOG ;;
    kb4nt)    KB_D=4 KB_ROUNDS=9 timeout 900 python3 tools/kbench.py t2.1.1088:16:0:0:0 t2.1.33856:16:0:0:0 t2.1.66624:16:0:0:0 t2.1.33872:16:0:0:0 x:16:0:0:0 t2.1.1088:16:0:1:0 t2.1.33856:16:0:1:0 t2.1.66624:16:0:1:0 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -24 ;;
    hostvis)  hipcc --offload-arch=gfx950 -O3 -w tools/hostvis_probe.hip -o /tmp/hostvis_probe >> $LOG 2>&1 && timeout 1500 /tmp/hostvis_probe ${ARG//,/ } >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -60 ;;
    gatherc)  timeout 300 ./sxxcvr_amd/lib/sx_gather_c ranks 1 0 /tmp/sx_rccl_id_$$ ${ARG:-22} 5 >> $LOG 2>&1 && timeout 300 ./sxxcvr_amd/lib/sx_gather_c all 1 ${ARG:-22} 5 >> $LOG 2>&1; RC=$?; tail -4 $LOG ;;
    kb4w)     # waves per CU capped through LDS padding (8 per CU, 32 generations) against 16 per CU, with and without nt loads; whole kernel and memory side; then the same on an all-zero input
              V="t2.1.1088:16:0:0:0:0 t2.1.66624:16:0:0:0:0 t2.1.1088:32:0:0:0:10600 t2.1.66624:32:0:0:0:10600 t2.1.66624:64:0:0:0:10600 t2.1.1088:16:0:1:0:0 t2.1.66624:16:0:1:0:0 t2.1.1088:32:0:1:0:10600 t2.1.66624:32:0:1:0:10600 t2.1.66624:64:0:1:0:0"
              KB_D=4 KB_ROUNDS=7 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?; KB_ZERO=1 KB_D=4 KB_ROUNDS=7 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped" ;;
    kb32h5)   # round 5: CF16 /32, the dense kernel with the typed LDS-DMA front end (x) against the multi-column kernel (w4), long interleaved visits, random and zero input
              KB_D=32 KB_FMT=CF16 KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 900 python3 tools/kbench.py x:8:0:0:0 w4:8:0:0:0 >> $LOG 2>&1; RC=$?
              KB_ZERO=1 KB_D=32 KB_FMT=CF16 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py x:8:0:0:0 w4:8:0:0:0 >> $LOG 2>&1
              grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped\|DIFFERENT\|rror" ;;
    valu5)    # round 5: the CF16 /32 mixes only (today's conversions in every reading lane against one conversion on the way into a CF32 image)
              hipcc --offload-arch=gfx950 -O3 -w tools/valu_power_probe.hip -o /tmp/valu_power_probe >> $LOG 2>&1 && timeout 300 /tmp/valu_power_probe from 22 >> $LOG 2>&1 && timeout 300 /tmp/valu_power_probe from 22 >> $LOG 2>&1; RC=$?; grep "ms per launch" $LOG ;;
    valu)     hipcc --offload-arch=gfx950 -O3 -w tools/valu_power_probe.hip -o /tmp/valu_power_probe >> $LOG 2>&1 && timeout 300 /tmp/valu_power_probe >> $LOG 2>&1; RC=$?; tail -22 $LOG ;;
    kb4x)     # one tile per wave (64 generations) with nt loads, taps by value, deferred stores, 12 waves per CU, pinned FMA order: whole kernel, then memory sides, then all-zero input
              V="x:16:0:0:0:0 t2.1.66624:16:0:0:0:0 t2.1.66624:32:0:0:0:0 t2.1.66624:64:0:0:0:0 t2.1.67136:64:0:0:0:0 t2.1.67136:16:0:0:0:0 t2.1.66625:16:0:0:0:0 t2.1.66624:21:0:0:0:3840 t2.1.132160:16:0:0:0:0 t2.1.197696:16:0:0:0:0"
              KB_D=4 KB_ROUNDS=7 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?
              KB_D=4 KB_ROUNDS=5 timeout 900 python3 tools/kbench.py t2.1.66624:32:0:1:0:0 t2.1.67136:64:0:1:0:0 t2.1.66625:16:0:1:0:0 t2.1.66624:21:0:1:0:3840 >> $LOG 2>&1
              KB_ZERO=1 KB_D=4 KB_ROUNDS=5 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped" ;;
    kbdnt)    for D in 32 8 16; do KB_D=$D KB_ROUNDS=7 timeout 600 python3 tools/kbench.py dense:8:0:0:0 densent:8:0:0:0 dense:8:0:1:0 densent:8:0:1:0 >> $LOG 2>&1; done; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped\|checksum" ;;
    kb4q)     # one persistent 16-wave workgroup per CU taking tiles from an LDS queue (one prologue per wave) with nt loads, against the shipped generations
              V="x:16:0:0:0:0 t2.16.66752:1:0:0:0:0 t2.16.1216:1:0:0:0:0 t2.8.66752:1:0:0:0:0 t2.4.66752:1:0:0:0:0 t2.16.66752:2:0:0:0:0 t2.1.66624:16:0:0:0:0"
              KB_D=4 KB_ROUNDS=7 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?
              KB_ZERO=1 KB_D=4 KB_ROUNDS=5 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped" ;;
    kb4f)     # fine tuning around the shipped /4 kernel (66624): generations, plain dispatch order, nt mask; 11 rounds, the shipped one first and last
              V="t2.1.66624:16:0:0:0:0 t2.1.66624:8:0:0:0:0 t2.1.66624:12:0:0:0:0 t2.1.66624:24:0:0:0:0 t2.1.66624:16:0:0:2:0 t2.1.263232:16:0:0:0:0 t2.1.67136:16:0:0:0:0 t2.1.66624:16:0:0:0:16"
              KB_D=4 KB_ROUNDS=11 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    kbnt2)    # nt masks that keep BOTH halos plain: /4 (525376) against the shipped 66624; dense /32, /8, /16: densent2 against densent / densepl
              KB_D=4 KB_ROUNDS=11 timeout 900 python3 tools/kbench.py t2.1.66624:16:0:0:0:0 t2.1.525376:16:0:0:0:0 t2.1.1088:16:0:0:0:0 t2.1.525376:16:0:1:0:0 t2.1.66624:16:0:1:0:0 t2.1.525376:16:0:0:0:16 >> $LOG 2>&1; RC=$?
              for D in 32 8 16; do KB_D=$D KB_ROUNDS=7 timeout 600 python3 tools/kbench.py densepl:8:0:0:0 densent:8:0:0:0 densent2:8:0:0:0 densent2:8:0:1:0 >> $LOG 2>&1; done; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    t2pmc)    # HBM traffic of /4 variants: FETCH_SIZE / WRITE_SIZE per launch (tools/pmc_pass.sh runs tools/onekernel.py)
              for V in t2:1:1088 t2:1:66624 t2:1:525376; do for C in FETCH_SIZE WRITE_SIZE; do SXFIR_TILE_VARIANT=$V SXFIR_PROF=1 bash tools/pmc_pass.sh "$C" 4 CF32 28 >> $LOG 2>&1; done; done; RC=$?; grep -v amdgpu.ids $LOG | tail -8 ;;
    kb4wide)  # eight outputs per lane (39.5 LDS reads per 512 FMAs, 2 waves per SIMD) with nt staging loads, against the shipped /4 kernel
              V="x:16:0:0:0:0 wident:16:0:0:0:0 wident12:16:0:0:0:0 wident16:16:0:0:0:0 wide:16:0:0:0:0 wident:8:0:0:0:0 wident:32:0:0:0:0 wident:16:0:1:0:0 x:16:0:0:0:16"
              KB_D=4 KB_ROUNDS=9 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?
              KB_ZERO=1 KB_D=4 KB_ROUNDS=5 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped" ;;
    kb4wide2) # the wide form's read-ahead depth and generations, nt loads, against the shipped kernel (first and last)
              V="x:16:0:0:0:0 wident16:16:0:0:0:0 wident24:16:0:0:0:0 wident32:16:0:0:0:0 wident16:8:0:0:0:0 wident16:32:0:0:0:0 wident16:4:0:0:0:0 wident12:16:0:0:0:0 x:16:0:0:0:16"
              KB_D=4 KB_ROUNDS=11 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    kb4wide3) # the shipped wide kernel (x) against round 3's form with nt loads (t2s), read-ahead depths and the pinned FMA order
              V="x:16:0:0:0:0 t2s:16:0:0:0:0 wident16:16:0:0:0:0 widentp24:16:0:0:0:0 widentp16:16:0:0:0:0 wident32:16:0:0:0:0 t2s:16:0:0:0:16 x:16:0:0:0:16"
              KB_D=4 KB_ROUNDS=11 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    kb4ab)    # wide (x) against round 3's form with nt loads (t2s): long runs (100 untimed + 200 timed launches per visit), 7 rounds, both orders
              KB_D=4 KB_ROUNDS=7 KB_ITERS=200 KB_SETTLE=100 timeout 1200 python3 tools/kbench.py x:16:0:0:0:0 t2s:16:0:0:0:0 wident16:16:0:0:0:0 wident32:16:0:0:0:0 widentp24:16:0:0:0:0 t2s:16:0:0:0:16 x:16:0:0:0:16 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    kb4wg)    # the shipped wide kernel: generations (oversub 8 / 16 / 32 / 64) and dispatch order, long visits
              KB_D=4 KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 1200 python3 tools/kbench.py x:16:0:0:0:0 x:8:0:0:0:0 x:32:0:0:0:0 x:64:0:0:0:0 x:16:0:0:2:0 x:12:0:0:0:0 x:16:0:0:0:16 >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    ib8p)     # x8: the scalar-tap pass kernel against the VGPR-tap tile kernel, generations, random and all-zero input
              timeout 900 python3 tools/ibench2.py pass:4 tile:4 pass4:4 pass:8 pass:16 pass:2 tile:4 pass:4 >> $LOG 2>&1; RC=$?; KB_ZERO=1 KB_ROUNDS=3 timeout 600 python3 tools/ibench2.py pass:4 tile:4 pass4:4 pass:16 >> $LOG 2>&1; grep -v amdgpu.ids $LOG | grep "ms med\|checksum\|all-zero" ;;
    kbdg)     # dense kernels (shipped nt mask): generations 4 / 8 / 16 / 32, long visits, /8 and /32
              for D in 8 32; do KB_D=$D KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 900 python3 tools/kbench.py dense:8:0:0:0 dense:4:0:0:0 dense:16:0:0:0 dense:32:0:0:0 dense:6:0:0:0 dense:8:0:0:0:16 >> $LOG 2>&1; done; RC=$?; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped" ;;
    kb8s)     # /8: the scalar-tap subset form against the shipped dense kernel, long visits; memory sides; all-zero input
              KB_D=8 KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 900 python3 tools/kbench.py densev:8:0:0:0 dense:8:0:0:0 dense:4:0:0:0 dense:16:0:0:0 densev:8:0:0:0:16 dense:8:0:0:0:16 >> $LOG 2>&1; RC=$?
              KB_ZERO=1 KB_D=8 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py dense:8:0:0:0 subset:8:0:0:0 >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped\|checksum\|all-zero" ;;
    pmc8s)    # LDS / VALU counters of the /8 subset form against the shipped dense kernel
              for V in 0 1; do SXFIR_DENSE_SUBSET=$V SXFIR_PROF=1 bash tools/pmc_pass.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" 8 CF32 28 >> $LOG 2>&1; done; RC=$?; grep -v amdgpu.ids $LOG | tail -3 ;;
    kbprev)   # before / after on one box: the profiling library of another commit (tools/prev_lib.sh) against this tree's, alternating
              # processes, long visits.  kbprev:D[,variant[,fmt]]  (variant default "dense:8:0:0:0")
              IFS=, read -r PD PV PF <<< "$ARG"; PD=${PD:-8}; PV=${PV:-dense:8:0:0:0}; PF=${PF:-CF32}
              echo "# previous = $(cat sxxcvr_amd/lib/prev/REV)" >> $LOG
              for R in 1 2 3; do
                echo "# previous" >> $LOG; SXFIR_PROF_LIB=$PWD/sxxcvr_amd/lib/prev/libsxfir_prof.so KB_FMT=$PF KB_D=$PD KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py $PV >> $LOG 2>&1 || RC=$?
                echo "# this tree" >> $LOG; KB_FMT=$PF KB_D=$PD KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py $PV >> $LOG 2>&1 || RC=$?
              done; grep -v "amdgpu.ids" $LOG | grep "^#\|ms med\|skipped" ;;
    ibprev)   # x8 interpolator, previous commit's profiling library against this tree's (as kbprev)
              echo "# previous = $(cat sxxcvr_amd/lib/prev/REV)" >> $LOG
              for R in 1 2 3; do
                echo "# previous" >> $LOG; SXFIR_PROF_LIB=$PWD/sxxcvr_amd/lib/prev/libsxfir_prof.so KB_ROUNDS=3 timeout 600 python3 tools/ibench2.py pass:8 >> $LOG 2>&1 || RC=$?
                echo "# this tree" >> $LOG; KB_ROUNDS=3 timeout 600 python3 tools/ibench2.py pass:8 >> $LOG 2>&1 || RC=$?
              done; grep -v "amdgpu.ids" $LOG | grep "^#\|ms med\|skipped" ;;
    kb4pol)   # /4 shipped kernel (x) against the same build with other cache policies on its nt loads / its stores (widepol<hex>:
              # low byte OR-ed into the loads' policy bits: 1 sc0, 10 sc1; next byte the stores: 1 plain, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1)
              V="x:16:0:0:0:0 widepol100:16:0:0:0:0 widepol200:16:0:0:0:0 widepol300:16:0:0:0:0 widepol400:16:0:0:0:0 widepol10:16:0:0:0:0 widepol11:16:0:0:0:0 widepol1:16:0:0:0:0 widepol310:16:0:0:0:0 widepol210:16:0:0:0:0 x:16:0:0:0:16"
              KB_D=4 KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 1500 python3 tools/kbench.py $V >> $LOG 2>&1; RC=$?
              KB_ZERO=1 KB_D=4 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 900 python3 tools/kbench.py $V >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | grep "ms med\|all-zero\|skipped\|DIFFERENT" ;;
    psplit4w) # power of the shipped /4 kernel, its memory side alone (nt loads) and round 3's form, beside the plain streams (mempower)
              PS_FORMS="wide whole=wident24:0,wide memory side=wident8:1,t2s whole=t2s:0,t2s memory side=t2.1.525376:1" timeout 300 python3 tools/power_split.py 4 >> $LOG 2>&1; RC=$?
              timeout 300 python3 tools/mempower.py >> $LOG 2>&1; grep -v "amdgpu.ids" $LOG | tail -12 ;;
    prices)   # instruction price list at the cap: the probe's mixes one at a time with the board's power sampled beside them
              hipcc --offload-arch=gfx950 -O3 -w tools/valu_power_probe.hip -o /tmp/valu_power_probe >> $LOG 2>&1 && timeout 600 python3 tools/price_list.py /tmp/valu_power_probe ${ARG:-2.5} >> $LOG 2>&1; RC=$?; grep -v amdgpu.ids $LOG | tail -16 ;;
    walls)    # per shipped kernel: random IQ (the cap), all-zero IQ (the structure), memory side alone -- long visits, one box
              for Z in 0 1; do
                echo "## KB_ZERO=$Z" >> $LOG
                KB_ZERO=$Z KB_D=4 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 400 python3 tools/kbench.py x:16:0:0:0 wident8:16:0:1:0 >> $LOG 2>&1 || RC=$?
                KB_ZERO=$Z KB_D=8 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 400 python3 tools/kbench.py dense:8:0:0:0 dense:8:0:1:0 >> $LOG 2>&1 || RC=$?
                KB_ZERO=$Z KB_D=32 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 400 python3 tools/kbench.py dense:8:0:0:0 dense:8:0:1:0 dense:8:0:2:0 >> $LOG 2>&1 || RC=$?
                KB_ZERO=$Z KB_D=32 KB_FMT=CF16 KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 400 python3 tools/kbench.py w4:8:0:0:0 w4:8:0:1:0 >> $LOG 2>&1 || RC=$?
                KB_ZERO=$Z KB_ROUNDS=3 timeout 400 python3 tools/ibench2.py pass:8 >> $LOG 2>&1 || RC=$?
              done; grep -v "amdgpu.ids" $LOG | grep "^##\|ms med\|skipped\|all-zero" ;;
    kbhc)     # dense /32 and /16 with halo carry (densehc) against the shipped form, whole kernel and memory side, random IQ and zeros
              for D in 32 16; do
                KB_D=$D KB_ROUNDS=5 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py dense:8:0:0:0 densehc:8:0:0:0 densehc:4:0:0:0 densehc:16:0:0:0 densehc:2:0:0:0 dense:8:0:0:0:16 >> $LOG 2>&1 || RC=$?
                KB_ZERO=1 KB_D=$D KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 600 python3 tools/kbench.py dense:8:0:0:0 densehc:8:0:0:0 dense:8:0:1:0 densehc:8:0:1:0 >> $LOG 2>&1 || RC=$?
              done; grep -v "amdgpu.ids" $LOG | grep "ms med\|skipped\|DIFFERENT\|all-zero" ;;
    kbq)      # the shipped /4 (and /8, /32) kernel on the same IQ quantised to 1 / 4 / 8 / 12 / 16 fractional bits, on zeros and at full precision: time against operand entropy
              for D in 4 32; do
                V=$([ $D = 4 ] && echo x:16:0:0:0 || echo dense:8:0:0:0)
                echo "## /$D" >> $LOG
                KB_ZERO=1 KB_D=$D KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 300 python3 tools/kbench.py $V >> $LOG 2>&1 || RC=$?
                for Q in 1 4 8 12 16; do KB_QBITS=$Q KB_D=$D KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 300 python3 tools/kbench.py $V >> $LOG 2>&1 || RC=$?; done
                KB_D=$D KB_ROUNDS=3 KB_ITERS=200 KB_SETTLE=100 timeout 300 python3 tools/kbench.py $V >> $LOG 2>&1 || RC=$?
              done; grep -v "amdgpu.ids" $LOG | grep "^##\|^# input\|all-zero\|ms med" ;;
    mb5)      hipcc --offload-arch=gfx950 -O3 -w tools/membench5.hip -o /tmp/membench5 >> $LOG 2>&1 && timeout 900 /tmp/membench5 "$ARG" >> $LOG 2>&1; RC=$?; grep -v "amdgpu.ids" $LOG | tail -150 ;;
    mb5pmc)   hipcc --offload-arch=gfx950 -O3 -w tools/membench5.hip -o /tmp/membench5 >> $LOG 2>&1 && bash tools/mb5_pmc.sh $OUT "$ARG" >> $LOG 2>&1; RC=$?; tail -60 $LOG ;;
    listpmc)  timeout 120 rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1; RC=$?; grep -c . $OUT/list_avail.txt ;;
    *)        echo "unknown step $S" | tee -a $LOG; RC=64 ;;
  esac
  echo "rc=$RC ($S)" | tee -a $LOG
  [ $RC -ne 0 ] && FAILED="$FAILED $S(rc=$RC)"
done
if [ -n "$FAILED" ]; then echo "FAILED STEPS:$FAILED"; exit 1; fi
exit 0
