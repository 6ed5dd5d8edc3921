#!/bin/bash
# threshold from group maxima: top-k parity + the bench's top-k object
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_round2.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r6m/pytest.txt 2>&1
tail -3 gpurun_out/r6m/pytest.txt
IMP_BENCH_DETAIL=gpurun_out/r6m/bench_detail.json python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6m/bench.line 2> gpurun_out/r6m/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6m/bench_detail.json'))['topk']
print({k:d[k] for k in ('value','knn_topk_recs_per_s','model_recommend_presliced_recs_per_s')})
print(d['kernels_ms_per_batch'])
PY
