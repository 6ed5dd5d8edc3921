#!/bin/bash
# quick GPU iteration: CG parity tests + C3 bench (+ optional extra shapes); usage: quick.sh <tag> [extra]
set -u
TAG=${1:-q}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
B="python bench.py --no-cpu-baseline --no-topk"
timeout 300 $B --steps 10 --warmup 2 > $O/c3.json 2> $O/c3.err
IMP_CG_STATS=1 timeout 200 $B --steps 1 --warmup 1 > /dev/null 2> $O/c3_stats.err
if [ "${2:-}" = "all" ]; then
timeout 300 $B --shape c2 --factors 64 --solver cg --steps 3 --warmup 1 > $O/c2_cg.json 2> $O/c2_cg.err
timeout 300 $B --shape ml20m --factors 128 --solver cg --steps 3 --warmup 1 > $O/c5_cg128.json 2> $O/c5_cg128.err
fi
tail -3 $O/tests.log
