#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for i in 1 2; do
IMP_BENCH_DETAIL=$O/base_$i.json $B > /dev/null 2>&1
for k in 12 24 48; do IMP_LIB_PATH=$R/build/variants/libimplicit_hip_rqst$k.so IMP_BENCH_DETAIL=$O/st${k}_$i.json $B > /dev/null 2>&1; done
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6l'
for f in sorted(glob.glob(O+'/*.json')):
    t=json.load(open(f))['topk']; k=t['kernels_ms_per_batch']
    print(os.path.basename(f), 'gemm %.4f'%k.get('score_gemm',0), 'knn %.0f'%t['knn_topk_recs_per_s'])
PY
