#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
B="timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-topk --no-extras"
IMP_BENCH_DETAIL=$O/base.json $B > /dev/null 2>&1
IMP_SHORT_SERVER=1 IMP_BENCH_DETAIL=$O/srv.json $B > /dev/null 2>&1
for k in 1 2 3; do IMP_SHORT_SERVER=1 IMP_LIB_PATH=$R/build/variants/libimplicit_hip_srvko$k.so IMP_BENCH_DETAIL=$O/srvko$k.json $B > /dev/null 2>&1; done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6i'
for f in sorted(glob.glob(O+'/*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print(os.path.basename(f), round(d['ms_per_step'],4), 'short', round(k.get('als_cg_short_rows',0),4))
PY
