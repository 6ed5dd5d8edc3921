#!/bin/bash
set -u
O=gpurun_out/${1:-topk}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu -k "topk or recommend or similar or checkerboard or config5" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/c3.json 2> $O/c3.err
IMP_TOPK_NO_EMIT=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/c3_noemit.json 2> $O/c3_noemit.err
python - <<PY
import json
for n in ("c3","c3_noemit"):
    d=json.load(open("$O/%s.json"%n))["topk"]
    print(n, round(d["value"]), round(d["scoring_TFLOPs"],1), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
