#!/bin/bash
# round 4: full -m gpu suite on the final tree + phase knock-outs of the blocked Cholesky (timing only)
set -u
TAG=${1:-r5s}; O=gpurun_out/$TAG; mkdir -p $O
(time timeout 2400 python -m pytest tests -q -m gpu) > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -8
cat > /tmp/chol128.py <<'PY'
import sys, warnings
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import numpy as np
import implicit_amd.gpu as gpu
import bench
from implicit_amd.synthetic import named
C = named("lastfm360k"); Ct = C.T.tocsr()
bench.FACTORS = int(sys.argv[2])
r = bench.extra_cholesky_f128(gpu, C, Ct)["cholesky_c3_f128"]
print(sys.argv[1], "f=%d" % bench.FACTORS, "ms/iter %.1f" % r["ms_per_iter"], "frac %.3f" % r["roofline"]["frac"], {k: round(v, 1) for k, v in r["kernels_ms_per_iter"].items()})
PY
for ko in 0 1 2 4 8 6 15; do IMP_CHOL_KO=$ko python /tmp/chol128.py ko$ko 128 2>&1 | grep "ms/iter"; done
