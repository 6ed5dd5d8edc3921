#!/bin/bash
# round 6: is it the 1024-thread workgroup (one gramian image per CU) or the team width?  team4 at 1024 threads (odd widths off),
# team3 at 512 threads (2 teams + 2 idle waves per workgroup, two workgroups per CU)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-topk --no-extras"
for i in 1 2; do
  IMP_TEAM_ODD=0 IMP_BENCH_DETAIL=$O/base_$i.json $B > /dev/null 2>&1
  IMP_TEAM_ODD=0 IMP_LIB_PATH=$R/build/variants/libimplicit_hip_t4b1024.so IMP_BENCH_DETAIL=$O/t4b1024_$i.json $B > /dev/null 2>&1
  IMP_LIB_PATH=$R/build/variants/libimplicit_hip_t3b512.so IMP_BENCH_DETAIL=$O/t3b512_$i.json $B > /dev/null 2>&1
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6c'
for f in sorted(glob.glob(f'{O}/*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print(os.path.basename(f), round(d['ms_per_step'],4), {n.replace('als_cg_','').replace('_rows',''):round(v,3) for n,v in k.items() if 'team' in n})
PY
