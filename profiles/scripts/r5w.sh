#!/bin/bash
# round 4: normal-matrix kernel after the atomics fix: parity, phase knock-outs, loop knock-outs (variant libraries)
set -u
TAG=${1:-r5w}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nm.py -q -m gpu -s > $O/nm_tests.log 2>&1; echo "nm tests rc=$?" >> $O/nm_tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/nm_tests.log | grep -E "per-row|passed|failed|FAILED|rc=|Error" | tail -25
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/b.json 2>/dev/null
  python - "$1" <<PY
import json, sys
j=json.loads(open("$O/b.json").read().strip().splitlines()[0])
print(sys.argv[1], "ms/step %.3f" % j["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in j["row_classes"].items()})
PY
}
for ko in 0 1 2 4 7; do IMP_NM_KO=$ko run "ko$ko"; done
for v in nomfma nosplit noboth; do IMP_LIB_PATH=build/variants/libimplicit_hip_$v.so run $v; done
IMP_NM=0 run old
