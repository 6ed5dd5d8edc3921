#!/bin/bash
mkdir -p gpurun_out/r6o
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_model.py tests/test_reference_suite.py -m gpu -x -q > gpurun_out/r6o/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r6o/pytest.txt | tail -5
python profiles/scripts/r6n_rechost.py 2>&1 | head -8
IMP_BENCH_DETAIL=gpurun_out/r6o/bench_detail.json python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6o/bench.line 2> gpurun_out/r6o/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6o/bench_detail.json'))['topk']
print({k:d[k] for k in ('value','knn_topk_recs_per_s','model_recommend_presliced_recs_per_s')})
PY
