#!/bin/bash
# round 4: why the packed-half tile kernel is slow -- fma_mix micro-rates + knock-outs of the kernel
set -u
TAG=${1:-r5j}; O=gpurun_out/$TAG; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate profiles/micro/valu_rate.hip 2>/dev/null && /tmp/valu_rate > $O/valu_rate.txt 2>&1
grep -i "mix\|cvt\|dot2" $O/valu_rate.txt
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
r = bench.extra_fp16(gpu, C, Ct, X0, Y0)["fp16_c3"]
print(sys.argv[1], "ms/iter %.3f" % r["ms_per_iter"], {k.replace("als_cg_", ""): round(v, 3) for k, v in r["kernels_ms_per_iter"].items() if "team" in k})
PY
export IMP_HALF_TILE64=1
python /tmp/fp16_time.py full 2>&1 | grep "ms/iter"
for v in qh_kotile qh_kodense qh_koboth; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so python /tmp/fp16_time.py $v 2>&1 | grep "ms/iter"; done
IMP_HALF_TILE64=0 python /tmp/fp16_time.py fp32tile 2>&1 | grep "ms/iter"
