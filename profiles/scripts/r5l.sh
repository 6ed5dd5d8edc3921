#!/bin/bash
# round 4: fp32 storage on 64-entry tiles ("fat" wavefronts): parity + per-class A/B
set -u
TAG=${1:-r5l}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "TILE64" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
for m in 0 15 8 12 0 15; do
  IMP_TILE64=$m timeout 300 $B > $O/b_m$m.json 2> $O/b_m$m.err
  python - <<PY
import json
j=json.load(open("$O/b_m$m.json"))
k=j["kernels_ms_per_step"]
print("mask $m ms/step %.3f" % j["ms_per_step"], {a.replace("als_cg_",""): round(v,3) for a,v in k.items() if "team" in a})
PY
done
