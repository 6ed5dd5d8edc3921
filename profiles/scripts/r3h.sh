#!/bin/bash
# round 3: per-phase cycle counters of the leader-protocol team kernels (IMP_CG_STATS), protocol tunables (leader priority, nap
# lengths) as A/B library builds, the top-k changes (fp16 read directly, sparse-candidate emit rule), Cholesky to f = 256
set -u
TAG=${1:-r3h}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
for v in prio2 prio3nap nap2 nap10; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err
done
timeout 300 $B > $O/b_base2.json 2> $O/b_base2.err
IMP_CG_STATS=1 timeout 300 python bench.py --no-cpu-baseline --no-topk --no-extras --steps 2 --warmup 1 > $O/stats.json 2> $O/cg_stats.err
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_als.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python - > $O/sim.txt 2>&1 <<'PY'
import sys, json, warnings
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import bench
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import SHAPES
out = bench.extra_c5(gpu, SHAPES)
print(json.dumps(out["similar_items_c5"]))
PY
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
grep -h "ms/step" $O/summary.txt; grep "cg-stats" $O/cg_stats.err | sort | uniq | head -12; tail -3 $O/tests.log; cat $O/sim.txt | cut -c1-900
