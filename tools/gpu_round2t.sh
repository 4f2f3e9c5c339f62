#!/bin/bash
# Round 2, GPU call T: whole GPU suite, driver-style bench, device-path probe after the DMA-copy pipeline.
set -u
OUT=gpurun_out/round2t
mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/round2t/bench.json") if x.startswith("{")][-1]
b=json.loads(l)
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["verified"])
print(json.dumps(b["through_device"], indent=1))
PY
(python3 tools/devpath_probe.py 1048576 40; python3 tools/devpath_probe.py 131072 40) 2>/dev/null > $OUT/devpath_probe.txt
cut -c1-160 $OUT/devpath_probe.txt
