#!/bin/bash
# top-k at factor counts off the 16-grid through zero-padded copies: parity + speed
set -u
TAG=${1:-r3y}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py tests/test_gpu_matrix.py "tests/test_gpu_round2.py::test_topk_emit_path_and_its_fallbacks" -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python -m pytest tests/test_reference_suite.py tests/test_gpu_model.py -x -q -m gpu > $O/tests_ref.log 2>&1; echo "tests_ref rc=$?" >> $O/tests_ref.log
timeout 250 python bench.py --factors 100 --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/f100.json 2> $O/f100.err
python - $O <<'PY'
import json, sys
j = json.load(open(sys.argv[1] + "/f100.json")); t = j["topk"]
print("f 100: iteration %.3f ms" % j["ms_per_step"], "topk recs/s %.0f" % t["value"], "recommend", t["model_recommend_recs_per_s"], {k: round(v, 3) for k, v in t["kernels_ms_per_batch"].items()})
PY
tail -3 $O/tests.log; tail -3 $O/tests_ref.log
