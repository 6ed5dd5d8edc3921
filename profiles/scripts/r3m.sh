#!/bin/bash
# top-k: scoring GEMM on the bf16 matrix cores with three-way split operands: parity (all top-k tests, reference suite) + A/B
set -u
TAG=${1:-r3m}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py "tests/test_gpu_round2.py::test_topk_emit_path_and_its_fallbacks" tests/test_gpu_fullsize.py::test_topk_full_item_count tests/test_gpu_fullsize.py::test_config5_f256_cg_and_similar_items_k100 -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
IMP_TOPK_FP32_MFMA=1 timeout 300 $B > $O/b0_fp32.json 2> $O/b0.err
timeout 300 $B > $O/b1_bf3.json 2> $O/b1.err
timeout 900 python -m pytest tests/test_reference_suite.py -x -q -m gpu > $O/tests_ref.log 2>&1; echo "tests_ref rc=$?" >> $O/tests_ref.log
python - <<'PY'
import json
for n in ("b0_fp32", "b1_bf3"):
    j = json.load(open(f"gpurun_out/r3m/{n}.json"))["topk"]
    print(n, "recs/s %.0f" % j["value"], "recommend %.0f" % j["model_recommend_recs_per_s"], {k: round(v, 4) for k, v in j["kernels_ms_per_batch"].items()}, j["roofline"]["achieved"], j["roofline"]["frac"])
PY
tail -3 $O/tests.log; tail -3 $O/tests_ref.log
