#!/bin/bash
# round 4: asm (compiler-invisible) rolling gathers with counted waits in the team kernels -- parity tests + A/B against the
# round-3 object on the same box
set -u
TAG=${1:-r5a}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_round2.py tests/test_gpu_golden.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for rep in 1 2; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_base.so timeout 300 $B > $O/b_base$rep.json 2> $O/b_base$rep.err
  timeout 300 $B > $O/b_new$rep.json 2> $O/b_new$rep.err
done
IMP_SHORT_STAGGER=0 timeout 300 $B > $O/b_new_nostagger.json 2> $O/b_new_nostagger.err
IMP_SHORT_STAGGER=0 IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_base.so timeout 300 $B > $O/b_base_nostagger.json 2> $O/b_base_nostagger.err
python profiles/scripts/show.py $O > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-400; tail -5 $O/tests.log
