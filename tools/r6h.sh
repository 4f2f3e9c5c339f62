#!/bin/bash
# round 6, third box: the bench line as the driver runs it, rocprofv3 + PMC summaries of every row that changed, CF16 / S32 size curves
set -u
mkdir -p gpurun_out/r6h
python3 bench.py > gpurun_out/r6h/bench_default.json 2> gpurun_out/r6h/bench_default.err
tail -c 600 gpurun_out/r6h/bench_default.json
bash tools/profile_round.sh round6 2 5 rx48 rx96 tx16 tx32 tx48 tx96 3tx 3rx > gpurun_out/r6h/profile.log 2>&1
tail -3 gpurun_out/r6h/profile.log
python3 tools/sizebench.py CF16 > gpurun_out/r6h/size_cf16.txt 2>&1
python3 tools/sizebench.py S32 > gpurun_out/r6h/size_s32.txt 2>&1
grep "relative" -A13 gpurun_out/r6h/size_cf16.txt gpurun_out/r6h/size_s32.txt
