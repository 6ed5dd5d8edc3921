#!/bin/bash
# tie rows resolved inside select_pruned: parity + per-kernel time of similar_items at configs[4]
mkdir -p gpurun_out/r6l
IMP_TOPK_DEBUG=1 python profiles/scripts/r6k_c5sim2.py > gpurun_out/r6l/c5sim.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_topk.py -m gpu -x -q > gpurun_out/r6l/pytest.txt 2>&1
tail -3 gpurun_out/r6l/pytest.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/r6l/bench.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
def find(o,key,path=''):
    if isinstance(o,dict):
        for k,v in o.items():
            if key in k: print(path+'/'+k, json.dumps(v)[:600])
            else: find(v,key,path+'/'+k)
find(d,'similar_items')
PY
