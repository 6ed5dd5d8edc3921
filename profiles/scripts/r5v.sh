#!/bin/bash
# round 4: phase knock-outs of the normal-matrix kernel (timing only)
set -u
TAG=${1:-r5v}; O=gpurun_out/$TAG; mkdir -p $O
for ko in 0 1 2 4 8 16 7 31; do
  IMP_NM_KO=$ko timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/b$ko.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("$O/b$ko.json").read().strip().splitlines()[0])
print("ko", $ko, "ms/step %.3f" % j["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in j["row_classes"].items()})
PY
done
