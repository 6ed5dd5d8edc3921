#!/bin/bash
# repro: the 16-wave team kernel at f = 64 on the C3 shape took seconds per launch
set -u
TAG=${1:-r3t}; O=gpurun_out/$TAG; mkdir -p $O
cat > /tmp/repro.py <<'PY'
import sys, time, warnings, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
C = named("lastfm360k")
lens = np.diff(C.indptr)
rows = np.nonzero((lens > 256) & (lens <= 512))[0]
f = int(sys.argv[1]); n = int(sys.argv[2])
sub = C[rows[:n]]
rng = np.random.default_rng(1)
X = gpu.Matrix(rng.random((sub.shape[0], f), dtype=np.float32) * 0.01)
Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
gram = gpu.Matrix.zeros(f, f)
s = gpu.LeastSquaresSolver(); s.calculate_yty(Y, gram, 0.01)
Cd = gpu.CSRMatrix(sub)
for rep in range(3):
    gpu.synchronize(); t0 = time.perf_counter(); s.least_squares(Cd, X, gram, Y, 3); gpu.synchronize()
    print(f"f={f} rows={sub.shape[0]} rep {rep}: {1e3*(time.perf_counter()-t0):.3f} ms", flush=True)
PY
for args in "64 64" "64 512" "64 513" "64 1024" "64 4561" "128 4561"; do
  timeout 120 python /tmp/repro.py $args >> $O/repro.txt 2>&1; echo "rc=$? ($args)" >> $O/repro.txt
done
IMP_TEAM_FUSED=0 timeout 120 python /tmp/repro.py 64 4561 >> $O/repro_old.txt 2>&1
cat $O/repro.txt $O/repro_old.txt
