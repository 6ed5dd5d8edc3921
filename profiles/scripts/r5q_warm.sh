#!/bin/bash
# round 5: the headline step against the number of warm-up steps (the device comes out of idle clocks when the bench starts)
set -u
O=gpurun_out/r5q; mkdir -p $O
for w in 5 25 100 400 5 25 100 400; do
  python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-topk --no-extras 2> /dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warmup', d['warmup'], 'ms_per_step %.4f' % d['ms_per_step'], 'events frac %.3f' % d['roofline']['frac_half_sweep_events'])"
done
