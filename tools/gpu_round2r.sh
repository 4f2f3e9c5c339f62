#!/bin/bash
# Round 2, GPU call R: device chains with DMA copies beside the kernels for large passes.
set -u
OUT=gpurun_out/round2r
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_device.py tests/test_gpu_bench.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/round2r/bench.json") if x.startswith("{")][-1]
b=json.loads(l)
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["verified"])
print(json.dumps(b["through_device"], indent=1))
PY
