#!/bin/bash
# ping-pong emit kernel (IMP_TOPK_PP=1): parity, then the bench's top-k object with and without
mkdir -p gpurun_out/r6p
IMP_TOPK_PP=1 timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r6p/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r6p/pytest.txt | tail -5
for pp in 0 1 0 1; do
IMP_TOPK_PP=$pp IMP_BENCH_DETAIL=gpurun_out/r6p/bench_pp$pp.json timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/r6p/bench_pp$pp.err
python - <<PY
import json
d=json.load(open('gpurun_out/r6p/bench_pp$pp.json'))['topk']
print('pp=$pp', {k:round(d[k]) for k in ('value','knn_topk_recs_per_s','model_recommend_presliced_recs_per_s')}, {k:round(v,4) for k,v in d['kernels_ms_per_batch'].items()})
PY
done
