#!/bin/bash
# round 5: timing-only knock-outs of the emit GEMM's fp16 form (build variants -DIMP_TOPK_KO=<mask>: 1 no MFMAs, 2 no item DMA, 4 no
# query DMA, 8 no low halves of the items, 16 no epilogue; results are wrong, only the score_gemm time is read)
set -u
for ko in ${KOS:-0 16 17 18 20 22 24 23 31 0}; do
  if [ $ko = 0 ]; then unset IMP_LIB_PATH; else export IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_topk_ko$ko.so; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2> /dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['topk']
k = d['kernels_ms_per_batch']
print('ko $ko', 'score_gemm %.4f' % k['score_gemm'], 'subset %.4f' % k['score_gemm_subset'], 'recs/s', round(d['value']))"
done
