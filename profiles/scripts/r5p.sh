#!/bin/bash
# round 4: f = 64 CG (configs[1] shape) on 64-entry tiles vs 32
set -u
TAG=${1:-r5p}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cg"
for m in 0 15 0 15 8 12; do
IMP_TILE64=$m timeout 600 $B > $O/c2_m$m.json 2> $O/c2_m$m.err
python - <<PY
import json
j=json.loads(open("$O/c2_m$m.json").read().strip().splitlines()[0]); print("mask $m ms/step %.3f" % j["ms_per_step"], {k.replace("als_cg_",""):round(v,2) for k,v in j["kernels_ms_per_step"].items() if "team" in k or "short" in k})
PY
done
