#!/bin/bash
# A/B of one environment switch on the C3 bench: ab_env.sh <tag> <VAR> <v1> <v2> ... ; optional EXTRA_ARGS env
set -u
O=gpurun_out/$1; VAR=$2; shift 2; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --steps 8 --warmup 2 ${EXTRA_ARGS:-}"
for v in "$@"; do
env $VAR=$v timeout 300 $B > $O/${VAR}_$v.json 2> $O/${VAR}_$v.err
done
