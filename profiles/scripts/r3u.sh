#!/bin/bash
# repro 2: C3 at f = 64, whole iterations, per-launch time of the (256,512] class, user / item side
set -u
TAG=${1:-r3u}; O=gpurun_out/$TAG; mkdir -p $O
cat > /tmp/repro2.py <<'PY'
import sys, time, warnings, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
C = named("lastfm360k"); Ct = C.T.tocsr()
f = 64
rng = np.random.default_rng(7)
X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
gram = gpu.Matrix.zeros(f, f)
s = gpu.LeastSquaresSolver()
Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
def t16():
    return gpu.Profiler.get("als_cg_team16_rows")
gpu.Profiler.reset(); gpu.Profiler.enable(True)
for it in range(int(sys.argv[1])):
    for side, (M, A, B) in enumerate(((Cd, X, Y), (Ctd, Y, X))):
        before = t16()[0]
        t0 = time.perf_counter()
        s.calculate_yty(B, gram, 0.01); s.least_squares(M, A, gram, B, 3); gpu.synchronize()
        a = A.to_numpy()
        print(f"iter {it} side {side}: wall {1e3*(time.perf_counter()-t0):9.2f} ms  team16 {t16()[0]-before:9.3f} ms  |A| max {np.abs(a).max():.3e} finite {np.isfinite(a).all()}", flush=True)
PY
timeout 200 python /tmp/repro2.py 10 > $O/new.txt 2>&1; echo "rc=$?" >> $O/new.txt
IMP_TEAM_FUSED=30 timeout 100 python /tmp/repro2.py 10 > $O/old_team16.txt 2>&1; echo "rc=$?" >> $O/old_team16.txt
cat $O/new.txt $O/old_team16.txt
