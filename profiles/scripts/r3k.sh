#!/bin/bash
# short rows: product-first and tile-first waves staggered between the same barriers: A/B + parity
set -u
TAG=${1:-r3k}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_SHORT_STAGGER=0 timeout 300 $B > $O/b0_seq.json 2> $O/b0.err
timeout 300 $B > $O/b1_stagger.json 2> $O/b1.err
IMP_SHORT_STAGGER=0 timeout 300 $B > $O/b2_seq.json 2> $O/b2.err
timeout 300 $B > $O/b3_stagger.json 2> $O/b3.err
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
grep -A2 "ms/step" $O/summary.txt | cut -c1-330; tail -3 $O/tests.log
