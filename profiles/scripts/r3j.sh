#!/bin/bash
# knock-out builds of the lock-step short-row kernel (timing only, results are wrong)
set -u
TAG=${1:-r3j}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 6 --warmup 2"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
for v in kg_mfma kg_tile kg_barrier kg_dots kg_mfma_tile kg_all; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3j/b_*.json")):
    try:
        j = json.load(open(f)); print(os.path.basename(f), "short_rows %.3f ms/iter" % j["kernels_ms_per_step"]["als_cg_short_rows"], "iteration %.3f" % j["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
