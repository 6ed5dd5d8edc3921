#!/bin/bash
# round 6: short rows on independent wavefronts with product servers (IMP_SHORT_G8=1) -- parity, then timing A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
IMP_SHORT_G8=1 timeout 600 python -m pytest tests/test_gpu_als.py -q -x -m gpu -k "warm_sweep and 128 or team_width or cold or edge" > $O/tests.log 2>&1; tail -4 $O/tests.log
B="timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-topk --no-extras"
for i in 1 2; do
  IMP_BENCH_DETAIL=$O/base_$i.json $B > /dev/null 2>&1
  IMP_SHORT_G8=1 IMP_BENCH_DETAIL=$O/g8_$i.json $B > /dev/null 2>$O/g8_$i.err
done
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6j'
for f in sorted(glob.glob(O+'/*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print(os.path.basename(f), round(d['ms_per_step'],4), 'short', round(k.get('als_cg_short_rows',0),4))
PY
