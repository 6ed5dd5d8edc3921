#!/bin/bash
O=gpurun_out/${1:-cholk}; mkdir -p $O
for k in 0 1 2 3; do
IMP_CHOL_KNOCK=$k timeout 300 python bench.py --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky --steps 2 --warmup 1 > $O/k$k.json 2> $O/k$k.err
python -c "
import json;d=json.load(open('$O/k$k.json'));print('knock $k: ms/iter', d['ms_per_step'])"
done
