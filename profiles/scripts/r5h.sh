#!/bin/bash
# round 4: float16 storage with packed 64-entry tiles (als_cg_qh.hip): parity + A/B against the fp32-tile kernels
set -u
TAG=${1:-r5h}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "fp16 or HALF_TILE or TEAM_FUSED=31" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -15
python - > $O/fp16.txt 2>&1 <<'PY'
import os, sys, time, warnings, json
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
r = bench.extra_fp16(gpu, C, Ct, X0, Y0)["fp16_c3"]
print("fp16_c3 ms/iter", r["ms_per_iter"], "frac", r["roofline"]["frac"])
print({k.replace("als_cg_", ""): round(v, 3) for k, v in r["kernels_ms_per_iter"].items()})
PY
cat $O/fp16.txt | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
IMP_HALF_TILE64=0 python - > $O/fp16_old.txt 2>&1 <<'PY'
import os, sys, time, warnings, json
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
r = bench.extra_fp16(gpu, C, Ct, X0, Y0)["fp16_c3"]
print("OLD fp16_c3 ms/iter", r["ms_per_iter"], "frac", r["roofline"]["frac"])
print({k.replace("als_cg_", ""): round(v, 3) for k, v in r["kernels_ms_per_iter"].items()})
PY
cat $O/fp16_old.txt | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
