#!/bin/bash
# emit pass: k-steps of item fragments in flight ahead of the products (RQ_AHEAD build variants), alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6q; mkdir -p $O; cd $R
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for a in 2 1 3 4; do
  L=$R/build/variants/libimplicit_hip_rqa$a.so; [ $a = 2 ] && L=$R/implicit_amd/libimplicit_hip.so
  IMP_LIB_PATH=$L IMP_BENCH_DETAIL=$O/a${a}_$rep.json $B > /dev/null 2>$O/a${a}_$rep.err
done; done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6q'
for f in sorted(glob.glob(O+'/*.json')):
    t=json.load(open(f))['topk']; k=t['kernels_ms_per_batch']
    print(os.path.basename(f), 'gemm %.4f'%k.get('score_gemm',0), 'knn %.0f'%t['knn_topk_recs_per_s'])
PY
