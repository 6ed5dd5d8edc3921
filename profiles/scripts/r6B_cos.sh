#!/bin/bash
mkdir -p gpurun_out/r6B
timeout 1500 python -m pytest tests/test_gpu_topk.py tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_reference_suite.py tests/test_gpu_logical_shards.py -m gpu -x -q > gpurun_out/r6B/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r6B/pytest.txt | tail -3
IMP_TOPK_DEBUG=1 IMP_BENCH_DETAIL=gpurun_out/r6B/bench_detail.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6B/bench.line 2> gpurun_out/r6B/bench.err
grep -c "topk-debug" gpurun_out/r6B/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6B/bench_detail.json'))
s=d['similar_items_c5']; print(s['items_per_s'], s['ms_per_batch'], s['kernels_ms_per_batch'])
t=d['topk']; print(t['value'], t['knn_topk_recs_per_s'])
PY
