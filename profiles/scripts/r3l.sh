#!/bin/bash
# short rows: gramian product on the bf16 matrix cores with three-way split operands (fp32-equivalent): A/B + parity
set -u
TAG=${1:-r3l}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_SHORT_BF16X3=0 timeout 300 $B > $O/b0_f32.json 2> $O/b0.err
timeout 300 $B > $O/b1_bf3.json 2> $O/b1.err
IMP_SHORT_BF16X3=0 timeout 300 $B > $O/b2_f32.json 2> $O/b2.err
timeout 300 $B > $O/b3_bf3.json 2> $O/b3.err
IMP_SHORT_STAGGER=0 timeout 300 $B > $O/b4_bf3_nostagger.json 2> $O/b4.err
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
grep -A2 "ms/step" $O/summary.txt | grep -v "^--" | cut -c1-200; grep -E "rel=|rel |lockstep|cold" $O/tests.log | head -30; tail -3 $O/tests.log
