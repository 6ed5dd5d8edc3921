#!/bin/bash
set -u
O=gpurun_out/${1:-ablock}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --steps 5 --warmup 1"
for k in 0 7; do
IMP_CG_LOCK=$k timeout 300 $B --shape c2 --factors 64 --solver cg > $O/c2_lock$k.json 2> $O/c2_lock$k.err
done
for k in 0 1; do
IMP_CG_LOCK=$k timeout 300 $B > $O/c3_lock$k.json 2> $O/c3_lock$k.err
done
