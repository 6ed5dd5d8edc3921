#!/bin/bash
# f = 64 Cholesky: panel sums on the matrix cores (default) against the all-vector form (IMP_CHOL_NO_PANEL_MFMA variant)
mkdir -p gpurun_out/r6A
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_model.py -m gpu -x -q -k "chol or golden or fold or recalc" > gpurun_out/r6A/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r6A/pytest.txt | tail -3
C="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky"
for rep in 1 2; do for v in panel; do
L=$PWD/implicit_amd/libimplicit_hip.so; [ $v = nopanel ] && L=$PWD/build/variants/libimplicit_hip_nopanel.so
IMP_LIB_PATH=$L IMP_BENCH_DETAIL=gpurun_out/r6A/$v.json $C > gpurun_out/r6A/$v.line 2> gpurun_out/r6A/$v.err
python -c "
import json; d=json.loads(open('gpurun_out/r6A/$v.line').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done; done
