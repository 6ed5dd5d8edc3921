#!/bin/bash
# round 4: team placement (waves of a team on one SIMD) and a lighter leader share -- A/B builds, timing + parity of each
set -u
TAG=${1:-r5b}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
timeout 300 $B > $O/b_0main.json 2> $O/b_0main.err
for v in samesimd light4 light8 ss_light4; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 600 python -m pytest tests/test_gpu_als.py -x -q -m gpu > $O/tests_$v.log 2>&1; echo "tests $v rc=$?" >> $O/tests_$v.log
done
timeout 300 $B > $O/b_1main.json 2> $O/b_1main.err
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
grep -A2 "ms/step" $O/summary.txt | cut -c1-330; grep -h "rc=" $O/tests_*.log
