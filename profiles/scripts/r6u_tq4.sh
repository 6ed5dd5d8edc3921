#!/bin/bash
# screened emit pass with four query tiles per wavefront (RQ_TQ4 variant) against two: parity of the top-k tests, then timing
mkdir -p gpurun_out/r6u
IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_tq4.so python - <<'PY' > gpurun_out/r6u/parity.txt 2>&1
import os, sys
sys.path.insert(0, '.')
import implicit_amd._libpath as lp
lp.OVERRIDE = os.environ['IMP_LIB_PATH']
import pytest
sys.exit(pytest.main(['tests/test_gpu_topk.py', '-m', 'gpu', '-x', '-q']))
PY
grep -n "passed\|failed\|rror" gpurun_out/r6u/parity.txt | tail -3
for v in base tq4 base tq4; do
L=$PWD/implicit_amd/libimplicit_hip.so; [ $v = tq4 ] && L=$PWD/build/variants/libimplicit_hip_tq4.so
IMP_LIB_PATH=$L IMP_BENCH_DETAIL=gpurun_out/r6u/$v.json python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r6u/$v.json'))['topk']
print('$v', round(d['knn_topk_recs_per_s']), {k:round(v,4) for k,v in d['kernels_ms_per_batch'].items() if k in ('score_gemm','topk_select_candidates')})
PY
done
