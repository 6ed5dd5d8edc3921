#!/bin/bash
# round 3: the new multi-GPU plumbing on one GPU (deferred sync, personalised exchange, shard-only fit), the parity additions
# (lockstep fit, trained-state full size, compiled reference on the box, ranking metrics + cross-loading), whole configs[3] on
# one GPU through the sharded driver, and two ranks on ONE device as a probe (RCCL may refuse it)
set -u
TAG=${1:-r3e}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sharded.py "tests/test_gpu_round2.py::test_model_fit_with_a_communicator" \
  "tests/test_gpu_model.py::test_lockstep_fit_every_half_sweep_f128" \
  "tests/test_reference_suite.py::test_ranking_metrics_and_cross_loading_with_stock_implicit" -x -q -m gpu -s > $O/tests_a.log 2>&1; echo "tests_a rc=$?" >> $O/tests_a.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s > $O/tests_full.log 2>&1; echo "tests_full rc=$?" >> $O/tests_full.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py --gpus 1 --shape c4 --steps 3 --warmup 1 > $O/c4_1gpu.json 2> $O/c4_1gpu.err; echo "c4 rc=$?" >> $O/c4_1gpu.err
# probe: two ranks on the same device
cat > /tmp/two.py <<'PY'
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.gpu import rendezvous
rank, world, local = rendezvous.env_world()
comm = rendezvous.init_comm(gpu, rank, world, 0)
m = gpu.Matrix(np.full((4, 4), rank + 1.0, dtype=np.float32))
comm.allreduce_sum(m)
print("rank", rank, "allreduce ->", m.to_numpy()[0, 0])
PY
for r in 0 1; do RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 120 python /tmp/two.py > $O/two_rank$r.log 2>&1 & done; wait
tail -3 $O/tests_a.log; tail -3 $O/tests_full.log; tail -2 $O/smoke.log; tail -2 $O/c4_1gpu.err; tail -3 $O/two_rank0.log
