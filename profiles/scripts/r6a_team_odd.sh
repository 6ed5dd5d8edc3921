#!/bin/bash
# round 6: team widths 3 / 5 (rows of (64,96] / (128,160]) against powers of two only, same box, alternating, 60-step runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_als.py -q -x -m gpu -k "team_width or warm_sweep" > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-topk --no-extras"
for i in 1 2 3; do
  IMP_BENCH_DETAIL=$O/odd1_$i.json $B > $O/odd1_$i.line 2>/dev/null
  IMP_TEAM_ODD=0 IMP_BENCH_DETAIL=$O/odd0_$i.json $B > $O/odd0_$i.line 2>/dev/null
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6b'
for tag in ('odd1','odd0'):
    for f in sorted(glob.glob(f'{O}/{tag}_*.json')):
        d=json.load(open(f))
        k=d['kernels_ms_per_step']
        print(tag, round(d['ms_per_step'],4), {n.replace('als_cg_','').replace('_rows',''):round(v,3) for n,v in k.items() if 'team' in n or 'short' in n})
PY
