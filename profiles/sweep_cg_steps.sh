#!/bin/bash
# Per-kernel time as a function of cg_steps (marginal cost of one CG pass vs the fixed gather + first-pass part).
for s in ${@:-0 1 2 3 6}; do
  IMP_BENCH_CG_STEPS=$s python bench.py --no-cpu-baseline --no-topk --steps 5 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('cg_steps=$s', round(d['ms_per_step'],3), {a[7:]:round(b,3) for a,b in k.items() if a.startswith('als_cg')})"
done
