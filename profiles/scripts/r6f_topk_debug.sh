#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
IMP_TOPK_DEBUG=1 IMP_BENCH_DETAIL=$O/detail.json timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/line.json 2> $O/bench.err
grep -c "topk-debug" $O/bench.err; grep "topk-debug" $O/bench.err | head -12
