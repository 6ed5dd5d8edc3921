#!/bin/bash
out=gpurun_out/${1:-r02f256}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_fullsize.py -q -m gpu -x -k "256 or config5 or config2_full" > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $out/tests.log | tail -4
for v in 0 1; do
  if [ $v = 1 ]; then export IMP_F256_GENERIC=1; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-topk --no-extras --shape ml20m --factors 256 > $out/c5_$v.json 2> $out/c5_$v.err
  python -c "
import json;d=json.load(open('$out/c5_$v.json'));print('generic=$v ms/iter', round(d['ms_per_step'],2), {k.replace('als_cg_',''):round(x,2) for k,x in d['kernels_ms_per_step'].items()})"
done
