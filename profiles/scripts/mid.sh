#!/bin/bash
# medium check: solver tests + round-2 tests, then the bench with extras
out=gpurun_out/${1:-r02mid}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_round2.py -q -m gpu -x > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log
tail -4 $out/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-topk > $out/bench.json 2> $out/bench.err
python profiles/scripts/show.py $out > $out/show.txt 2>&1
head -4 $out/show.txt
