#!/bin/bash
# round 3: the rolling-gather short-row kernel (IMP_TEAM_FUSED bit 32): parity, A/B against the round-2 lock-step kernel,
# and the VALU issue-rate micro-benchmark with its placement check
set -u
TAG=${1:-r3d}; O=gpurun_out/$TAG; mkdir -p $O
timeout 120 ./build/valu_rate > $O/valu_rate.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_TEAM_FUSED=31 timeout 300 $B > $O/b0_old.json 2> $O/b0.err
timeout 300 $B > $O/b1_new.json 2> $O/b1.err
IMP_QGROUP_PER_CU=1 timeout 300 $B > $O/b2_new_percu1.json 2> $O/b2.err
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
cat $O/valu_rate.txt $O/summary.txt
