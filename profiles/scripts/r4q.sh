#!/bin/bash
# top-k: pinned staging of flags + results (one host wait per call): parity and the call rate
set -u
O=gpurun_out/${1:-r4q}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout 300 $B > $O/c3.json 2> $O/c3.err
python - <<PY
import json
d=json.load(open("$O/c3.json"))["topk"]
print(round(d["value"]), round(d.get("model_recommend_recs_per_s", 0)), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
