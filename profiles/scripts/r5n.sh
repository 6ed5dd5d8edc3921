#!/bin/bash
# round 4: switch tests after the tile64 / team16-cluster fix, the N > 1 bench path with one rank, f = 192 padded vs generic
set -u
TAG=${1:-r5n}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -6
IMP_FORCE_SHARDED=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scale 0.1 --shape c4 > $O/sharded.json 2> $O/sharded.err
python - <<PY
import json
j=json.loads(open("$O/sharded.json").read().strip().splitlines()[0])
print("sharded 1 rank x0.1: ms/step %.2f" % j["ms_per_step"], "compute", round(j["rank0_compute_ms_per_step"],2), "exposed", round(j["rank0_exposed_exchange_ms_per_step"],2), j["roofline"] and round(j["roofline"]["frac"],3), j["exchange_schemes_GB_per_rank_per_step"])
PY
cat > /tmp/f192.py <<'PY'
import sys, warnings
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import numpy as np
import implicit_amd.gpu as gpu
import bench
from implicit_amd.synthetic import named
C = named("lastfm360k"); Ct = C.T.tocsr()
f = 192
rng = np.random.default_rng(7)
X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
gram = gpu.Matrix.zeros(f, f)
solver = gpu.LeastSquaresSolver()
def cg():
    solver.calculate_yty(Y, gram, 0.01); solver.least_squares(Cd, X, gram, Y, 3)
    solver.calculate_yty(X, gram, 0.01); solver.least_squares(Ctd, Y, gram, X, 3)
t, k = bench._time_iterations(gpu, cg)
print(sys.argv[1], "f=192 ms/iter %.2f" % (1e3 * t), {a: round(b, 2) for a, b in k.items()})
PY
python /tmp/f192.py padded 2>&1 | grep "ms/iter"
IMP_NO_PAD=1 python /tmp/f192.py generic 2>&1 | grep "ms/iter"
