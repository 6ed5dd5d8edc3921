#!/bin/bash
# round 6: knock-outs of the resident scoring kernel (timing only): RQ_KO bits 1 no MFMAs, 2 no item DMA, 4 no epilogue, 8 no barrier
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
IMP_BENCH_DETAIL=$O/base.json $B > /dev/null 2>$O/base.err
for k in ; do
  IMP_LIB_PATH=$R/build/variants/libimplicit_hip_rqko$k.so IMP_BENCH_DETAIL=$O/ko$k.json $B > /dev/null 2>$O/ko$k.err
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6f'
for f in sorted(glob.glob(O+'/*.json')):
    t=json.load(open(f))['topk']; k=t['kernels_ms_per_batch']
    print(os.path.basename(f), 'gemm %.4f subset %.4f'%(k.get('score_gemm',0),k.get('score_gemm_subset',0)), 'knn %.0f'%t['knn_topk_recs_per_s'])
PY
