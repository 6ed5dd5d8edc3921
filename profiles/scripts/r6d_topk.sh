#!/bin/bash
# round 6: resident-query scoring kernel -- top-k tests, then the bench's top-k leg (kernel times per batch)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -x -m gpu -k "topk or emit or fp16_form or golden or similar or recommend or plane or c5 or switch" > $O/tests.log 2>&1; tail -5 $O/tests.log
IMP_BENCH_DETAIL=$O/detail.json timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/line.json 2> $O/bench.err
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6e'
d=json.load(open(O+'/detail.json')); t=d['topk']
print('recommend', t['value'], 'knn', t['knn_topk_recs_per_s']); print(t['kernels_ms_per_batch']); print(t['roofline']['achieved'], t['roofline']['frac'])
s=d.get('similar_items_c5'); print(s and (s['items_per_s'], s['kernels_ms_per_batch'], s['roofline']))
print(open(O+'/line.json').read()[:3000])
PY
