#!/bin/bash
# top-k GEMM, 16x16x32 split-bf16 form: occupancy variants + parity
set -u
TAG=${1:-r3o}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
timeout 300 $B > $O/b_w3.json 2> $O/b_w3.err
for v in tkw1 tkw2; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err; done
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py "tests/test_gpu_round2.py::test_topk_emit_path_and_its_fallbacks" tests/test_gpu_fullsize.py::test_topk_full_item_count tests/test_gpu_fullsize.py::test_config5_f256_cg_and_similar_items_k100 -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3o/b_*.json")):
    j = json.load(open(f))["topk"]
    print(os.path.basename(f), "recs/s %.0f" % j["value"], "recommend %.0f" % j["model_recommend_recs_per_s"], "gemm ms %.4f" % j["kernels_ms_per_batch"]["score_gemm"])
PY
tail -3 $O/tests.log
