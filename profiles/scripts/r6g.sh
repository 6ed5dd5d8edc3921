#!/bin/bash
# round 4: blocked Cholesky after the panel / back-substitution changes: parity + timing (f = 128, 100, 256), packed A/B
set -u
TAG=${1:-r6g}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_round2.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q -m gpu -k "chol or golden or fold or CHOL or recalculate or partial_fit" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -5
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
print(sys.argv[1], "f=%d" % bench.FACTORS, "ms/iter %.1f" % r["ms_per_iter"], "frac %.3f" % r["roofline"]["frac"])
PY
python /tmp/chol128.py blocked 128 2>&1 | grep "ms/iter"
IMP_CHOL_PACKED=128 python /tmp/chol128.py packed 128 2>&1 | grep "ms/iter"
for ko in 2 4 8 15; do IMP_CHOL_KO=$ko python /tmp/chol128.py ko$ko 128 2>&1 | grep "ms/iter"; done
python /tmp/chol128.py blocked 100 2>&1 | grep "ms/iter"
python /tmp/chol128.py blocked 256 2>&1 | grep "ms/iter"
