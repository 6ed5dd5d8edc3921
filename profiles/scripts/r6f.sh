#!/bin/bash
# round 4: normal-matrix kernel: segment size and wave priority A/B
set -u
TAG=${1:-r6f}; O=gpurun_out/$TAG; mkdir -p $O
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/b.json 2>/dev/null
  python - "$1" <<PY
import json, sys
j=json.loads(open("$O/b.json").read().strip().splitlines()[0])
print(sys.argv[1], "ms/step %.3f" % j["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in j["row_classes"].items()})
PY
}
run base
IMP_LIB_PATH=build/variants/libimplicit_hip_setprio.so run setprio
IMP_NM_SEGMENT=1024 run seg1024
IMP_NM_SEGMENT=4096 run seg4096
IMP_NM_SEGMENT=8192 run seg8192
run base2
