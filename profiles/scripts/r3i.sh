#!/bin/bash
# knock-out builds of the leader-protocol team kernels (timing only, results are wrong): what a pass is made of
set -u
TAG=${1:-r3i}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 6 --warmup 2"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
for v in ko_dread ko_dfma ko_dense ko_tile ko_sync ko_dense_tile ko_all; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err
done
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
IMP_CSR_TIMING=1 python - > $O/csr_timing.txt 2>&1 <<'PY'
import sys, time, warnings
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import SHAPES, grid_shards
users, items, nnz, gamma = SHAPES["c4"]
t0 = time.time(); Cui, Ciu, _, _ = grid_shards(0, 1, users, items, nnz, 8, gamma=gamma, seed=42); print("generate", time.time() - t0)
for name, M in (("Cui", Cui), ("Ciu", Ciu)):
    t0 = time.time(); d = gpu.CSRMatrix(M); print(name, "CSRMatrix total", time.time() - t0, flush=True); del d
PY
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
grep -A1 "ms/step" $O/summary.txt | cut -c1-420; tail -3 $O/tests.log; cat $O/csr_timing.txt
