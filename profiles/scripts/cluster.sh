#!/bin/bash
# cluster-resident long rows: parity tests, then the bench line with the secondary objects
out=gpurun_out/${1:-r02cl}
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -k "cluster or NO_CLUSTER or fp16" > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-topk > $out/bench.json 2> $out/bench.err
python profiles/scripts/show.py $out > $out/show.txt 2>&1
tail -c 1500 $out/show.txt
