#!/bin/bash
# top-k emit epilogue trimmed (thresholds by 16-byte loads, float pre-test): timing + parity
set -u
TAG=${1:-r3r}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
for i in 1 2; do timeout 300 $B > $O/b_$i.json 2> $O/b_$i.err; done
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py "tests/test_gpu_round2.py::test_topk_emit_path_and_its_fallbacks" tests/test_gpu_fullsize.py::test_topk_full_item_count -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python - $O <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    j = json.load(open(f))["topk"]; print(os.path.basename(f), "recs/s %.0f" % j["value"], "recommend %.0f" % j["model_recommend_recs_per_s"], {k: round(v, 4) for k, v in j["kernels_ms_per_batch"].items()})
PY
tail -2 $O/tests.log
