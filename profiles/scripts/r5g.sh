#!/bin/bash
# round 4: cluster fault path (repaired on the device), regression check of the cluster kernels
set -u
TAG=${1:-r5g}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "cluster or lost or oversub" -rP > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -25
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 10 --warmup 3"
timeout 300 $B > $O/b_new.json 2> $O/b_new.err
IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_base.so timeout 300 $B > $O/b_base.json 2> $O/b_base.err
python profiles/scripts/show.py $O 2>&1 | grep -A2 "ms/step" | cut -c1-400
