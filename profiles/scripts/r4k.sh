#!/bin/bash
# top-k: fragment-ordered pre-split query rows: parity (top-k + golden) and A/B
set -u
O=gpurun_out/${1:-r4k}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout 300 $B > $O/c3.json 2> $O/c3.err
IMP_TOPK_NO_QSPLIT=1 timeout 300 $B > $O/c3_noqsplit.json 2> $O/c3_noqsplit.err
python - <<PY
import json
for n in ("c3","c3_noqsplit"):
    d=json.load(open("$O/%s.json"%n))["topk"]
    print(n, round(d["value"]), round(d["scoring_TFLOPs"],1), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
