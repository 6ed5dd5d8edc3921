#!/bin/bash
# round-6 verification run: whole -m gpu suite, smoke, driver-style bench line
mkdir -p gpurun_out/r6z
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6z/pytest.txt 2>&1
tail -3 gpurun_out/r6z/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z/smoke.txt 2>&1; echo "smoke rc=$?"
IMP_BENCH_DETAIL=gpurun_out/r6z/bench_detail.json python bench.py --steps 20 --warmup 5 > gpurun_out/r6z/bench.line 2> gpurun_out/r6z/bench.err
wc -c gpurun_out/r6z/bench.line
