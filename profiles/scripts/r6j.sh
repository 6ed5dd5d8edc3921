#!/bin/bash
# round 4: normal-matrix kernel, staggered schedule (variant library) against the two-phase rounds
set -u
TAG=${1:-r6j}; O=gpurun_out/$TAG; mkdir -p $O
IMP_LIB_PATH=build/variants/libimplicit_hip_stagger.so timeout 600 python -m pytest tests/test_gpu_nm.py -q -m gpu -k "not old_long" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/b.json 2>/dev/null
  python - "$1" <<PY
import json, sys
j=json.loads(open("$O/b.json").read().strip().splitlines()[0])
print(sys.argv[1], "ms/step %.3f" % j["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in j["row_classes"].items()})
PY
}
run base
IMP_LIB_PATH=build/variants/libimplicit_hip_stagger.so run stagger
run base2
IMP_LIB_PATH=build/variants/libimplicit_hip_stagger.so run stagger2
