#!/bin/bash
# knock-out builds of the split-bf16 scoring GEMM (timing only): operand loads, epilogue, matrix work
set -u
TAG=${1:-r3q}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
for v in tk_l1 tk_noload tk_noepi tk_nomfma tk_noload_noepi; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err; done
python - $O <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        j = json.load(open(f))["topk"]; print(os.path.basename(f), "gemm ms %.4f" % j["kernels_ms_per_batch"]["score_gemm"], "subset %.4f" % j["kernels_ms_per_batch"]["score_gemm_subset"])
    except Exception as e: print(f, e)
PY
