#!/bin/bash
# round-6 wrap-up on one box: collect.sh r06 (profiles), PARITY.md, whole -m gpu suite, smoke, driver-style bench line
mkdir -p gpurun_out/r6w
bash profiles/collect.sh r06 > gpurun_out/r6w/collect.log 2>&1
python profiles/parity_report.py > gpurun_out/r6w/parity.log 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6w/pytest.txt 2>&1
grep -n "passed\|failed" gpurun_out/r6w/pytest.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6w/smoke.txt 2>&1; echo "smoke rc=$?"
IMP_BENCH_DETAIL=gpurun_out/r6w/bench_detail.json python bench.py --steps 20 --warmup 5 > gpurun_out/r6w/bench.line 2> gpurun_out/r6w/bench.err
wc -c gpurun_out/r6w/bench.line
