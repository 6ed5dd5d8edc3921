#!/bin/bash
# top-k GEMM (split-bf16 form): waves per SIMD the register allocation leaves room for
set -u
TAG=${1:-r3n}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
timeout 300 $B > $O/b_w2.json 2> $O/b_w2.err
for v in tkw3 tkw4; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err; done
IMP_TOPK_FP32_MFMA=1 IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_tkw3.so timeout 300 $B > $O/b_tkw3_fp32.json 2> $O/b_tkw3_fp32.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3n/b_*.json")):
    j = json.load(open(f))["topk"]
    print(os.path.basename(f), "recs/s %.0f" % j["value"], "recommend %.0f" % j["model_recommend_recs_per_s"], "gemm ms %.4f" % j["kernels_ms_per_batch"]["score_gemm"])
PY
