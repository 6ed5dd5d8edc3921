#!/bin/bash
set -u
TAG=${1:-r5k}; O=gpurun_out/$TAG; mkdir -p $O
cat > /tmp/fp16_time.py <<'PY'
import os, sys, warnings
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import numpy as np
import implicit_amd.gpu as gpu
import bench
from implicit_amd.synthetic import named
C = named("lastfm360k"); Ct = C.T.tocsr()
rng = np.random.default_rng(7)
X0 = rng.random((C.shape[0], 128), dtype=np.float32) * 0.01
Y0 = rng.random((C.shape[1], 128), dtype=np.float32) * 0.01
for rep in range(2):
    r = bench.extra_fp16(gpu, C, Ct, X0, Y0)["fp16_c3"]
    print(sys.argv[1], rep, "ms/iter %.3f" % r["ms_per_iter"], {k.replace("als_cg_", ""): round(v, 3) for k, v in r["kernels_ms_per_iter"].items() if "team" in k or "short" in k})
PY
IMP_HALF_TILE64=1 python /tmp/fp16_time.py tile64 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
IMP_HALF_TILE64=0 python /tmp/fp16_time.py fp32tile 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
IMP_HALF_TILE64=1 python /tmp/fp16_time.py tile64 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
IMP_HALF_TILE64=1 timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "fp16 or HALF_TILE" 2>&1 | tail -30
