#!/bin/bash
# stride of the threshold pre-pass's subset (RQ_SUBSTRIDE variants): every 32nd / 64th / 128th 128-item block
mkdir -p gpurun_out/r6x
for rep in 1 2; do for v in 32 64 128; do
L=$PWD/implicit_amd/libimplicit_hip.so; [ $v != 32 ] && L=$PWD/build/variants/libimplicit_hip_ss$v.so
IMP_TOPK_DEBUG=1 IMP_LIB_PATH=$L IMP_BENCH_DETAIL=gpurun_out/r6x/s$v.json python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/r6x/s$v.err
python - <<PY
import json
d=json.load(open('gpurun_out/r6x/s$v.json'))['topk']
print('stride $v', round(d['knn_topk_recs_per_s']), round(d['value']), {k:round(x,4) for k,x in d['kernels_ms_per_batch'].items()}, 'fallback lines', open('gpurun_out/r6x/s$v.err').read().count('topk-debug'))
PY
done; done
