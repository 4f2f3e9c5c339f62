#!/bin/bash
# Round 2, GPU call N: multi-column decimator variants at /32 and /8, interpolator oversub; device tests.
set -u
OUT=gpurun_out/round2n
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_device.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export KB_ROUNDS=9 KB_ITERS=20
KB_D=32 timeout 300 python3 tools/kbench.py w4:8:0:0:0 w8:8:0:0:0 w4:4:0:0:0 w4:16:0:0:0 w4:2:0:0:0 w8:4:0:0:0 w4:8:0:1:0 w4:8:0:2:0 > $OUT/kbench_d32.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_d32.txt
KB_D=8 timeout 300 python3 tools/kbench.py w4:8:0:0:0 w2:8:0:0:0 w1:8:0:0:0 w4:16:0:0:0 w4:4:0:0:0 w4:8:0:1:0 w4:8:0:2:0 > $OUT/kbench_d8.txt 2>&1
grep -v "amdgpu.ids\|checksum" $OUT/kbench_d8.txt
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['through_device'], indent=0)[:1500])"
