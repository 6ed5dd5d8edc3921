#!/bin/bash
# top-k GEMM: 3 vs 4 waves per SIMD (workgroup-shared staging leaves 128 registers enough), default bench line
set -u
O=gpurun_out/${1:-r4p}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout 300 $B > $O/c3.json 2> $O/c3.err
IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_w4.so timeout 300 $B > $O/c3_w4.json 2> $O/c3_w4.err
python - <<PY
import json
for n in ("c3","c3_w4"):
    d=json.load(open("$O/%s.json"%n))["topk"]
    print(n, round(d["value"]), round(d.get("model_recommend_recs_per_s", 0)), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
timeout 200 python profiles/scripts/topk_kernels.py 2>&1 | tail -1
IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_w4.so timeout 200 python profiles/scripts/topk_kernels.py 2>&1 | tail -1
