#!/bin/bash
# screened emit pass (MODE 3, default) against the three-product pass (IMP_TOPK_SCREEN=0): parity, then the bench's top-k object
mkdir -p gpurun_out/r6s
timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_reference_suite.py -m gpu -x -q > gpurun_out/r6s/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r6s/pytest.txt | tail -5
for sc in 1 0 1 0; do
IMP_TOPK_SCREEN=$sc IMP_TOPK_DEBUG=1 IMP_BENCH_DETAIL=gpurun_out/r6s/bench_s$sc.json timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/r6s/bench_s$sc.err
grep -c "topk-debug" gpurun_out/r6s/bench_s$sc.err
python - <<PY
import json
d=json.load(open('gpurun_out/r6s/bench_s$sc.json'))['topk']
print('screen=$sc', {k:round(d[k]) for k in ('value','knn_topk_recs_per_s','model_recommend_presliced_recs_per_s')}, {k:round(v,4) for k,v in d['kernels_ms_per_batch'].items()})
PY
done
