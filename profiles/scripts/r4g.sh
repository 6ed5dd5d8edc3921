#!/bin/bash
# short rows: two independent half-groups of 8 rows per workgroup (own barriers): A/B + parity
set -u
TAG=${1:-r4g}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_SHORT_HALVES=0 timeout 300 $B > $O/b0_group16.json 2> $O/b0.err
timeout 300 $B > $O/b1_halves.json 2> $O/b1.err
IMP_SHORT_HALVES=0 timeout 300 $B > $O/b2_group16.json 2> $O/b2.err
timeout 300 $B > $O/b3_halves.json 2> $O/b3.err
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python - $O <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/b*.json")):
    j = json.load(open(f)); print(os.path.basename(f), "ms/iter %.3f" % j["ms_per_step"], "short %.3f" % j["kernels_ms_per_step"]["als_cg_short_rows"])
PY
tail -3 $O/tests.log
