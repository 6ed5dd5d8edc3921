#!/bin/bash
# round 4: the row classes of a half sweep on four streams (IMP_CLASS_STREAMS=1): parity subset + A/B of the bench line
set -u
TAG=${1:-r6l}; O=gpurun_out/$TAG; mkdir -p $O
run() {
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-topk > $O/b.json 2>/dev/null
  python - "$1" <<PY
import json, sys
j=json.loads(open("$O/b.json").read().strip().splitlines()[0])
print(sys.argv[1], "ms/step %.3f" % j["ms_per_step"], "half-sweep events %.3f" % j["roofline"]["avg_launch_ms"])
PY
}
run base
IMP_CLASS_STREAMS=1 run streams
run base2
IMP_CLASS_STREAMS=1 run streams2
IMP_CLASS_STREAMS=1 timeout 600 python -m pytest tests/test_gpu_nm.py tests/test_gpu_als.py tests/test_gpu_golden.py -q -m gpu -x -k "not chol and not gramian and not loss" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
