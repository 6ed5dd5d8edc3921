#!/bin/bash
# top-k threshold pass: k rounds of block arg-max (small k) instead of the 4-pass radix select: parity + batch time
set -u
O=gpurun_out/${1:-r4h}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/c3.json 2> $O/c3.err
python - <<PY
import json
d=json.load(open("$O/c3.json"))["topk"]
print(round(d["value"]), round(d["scoring_TFLOPs"],1), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
