#!/bin/bash
# round 3: parity of the leader-protocol team kernels + A/B bench against the round-2 kernels (same box, same call)
set -u
TAG=${1:-r3b}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
IMP_TEAM_FUSED=0 timeout 300 $B > $O/b0_old.json 2> $O/b0.err
timeout 300 $B > $O/b1_new.json 2> $O/b1.err
IMP_TEAM_FUSED=0 timeout 300 $B --shape c2 --factors 64 --solver cg --steps 4 --warmup 1 > $O/c2_old.json 2> $O/c2_old.err
timeout 300 $B --shape c2 --factors 64 --solver cg --steps 4 --warmup 1 > $O/c2_new.json 2> $O/c2_new.err
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
