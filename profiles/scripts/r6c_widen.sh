#!/bin/bash
# round 6: each mid class on teams of twice the wavefronts (half the share per wavefront, half the rows in flight)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-topk --no-extras"
for m in 0 8 4 2 14; do
  IMP_TEAM_WIDEN=$m IMP_BENCH_DETAIL=$O/widen_$m.json $B > /dev/null 2>&1
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6d'
for f in sorted(glob.glob(f'{O}/*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print(os.path.basename(f), round(d['ms_per_step'],4), {n.replace('als_cg_','').replace('_rows',''):round(v,3) for n,v in k.items() if 'team' in n})
PY
