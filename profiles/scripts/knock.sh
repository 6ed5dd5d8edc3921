#!/bin/bash
set -u
O=gpurun_out/${1:-knock}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --steps 6 --warmup 2"
for k in 0 1 2 4 8 3 7 15; do
IMP_CG_KNOCK=$k timeout 200 $B > $O/k$k.json 2> $O/k$k.err
done
