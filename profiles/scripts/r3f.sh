#!/bin/bash
# round 3: short-row kernel with the product and the tile entries interleaved (A/B + parity), and where fit()'s set-up time goes
set -u
TAG=${1:-r3f}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_SHORT_OVERLAP=0 timeout 300 $B > $O/b0_seq.json 2> $O/b0.err
timeout 300 $B > $O/b1_overlap.json 2> $O/b1.err
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python - > $O/fit_setup.txt 2>&1 <<'PY'
import sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
from implicit_amd.utils import check_csr, check_random_state
C = named("lastfm360k")
def t(label, fn):
    t0 = time.perf_counter(); r = fn(); gpu.synchronize(); print(f"{label:40s} {1e3*(time.perf_counter()-t0):8.1f} ms"); return r
for rep in range(2):
    print("rep", rep)
    Cui = t("check_csr", lambda: check_csr(C))
    Ciu = t("Cui.T.tocsr()", lambda: Cui.T.tocsr())
    rng = check_random_state(1)
    x0 = t("rng.random users*0.01", lambda: rng.random((C.shape[0], 128), dtype=np.float32) * 0.01)
    y0 = t("rng.random items*0.01", lambda: rng.random((C.shape[1], 128), dtype=np.float32) * 0.01)
    X = t("upload X", lambda: gpu.Matrix(x0)); Y = t("upload Y", lambda: gpu.Matrix(y0))
    Cd = t("CSRMatrix(Cui)", lambda: gpu.CSRMatrix(Cui)); Ctd = t("CSRMatrix(Ciu)", lambda: gpu.CSRMatrix(Ciu))
    Xd = t("device uniform users", lambda: gpu.RandomState(1).uniform(C.shape[0], 128, 0.0, 0.01))
PY
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
cat $O/summary.txt $O/fit_setup.txt
