#!/bin/bash
# round-2 call 1: MFMA 4x4x1 probe, baselines of every BASELINE shape with the round-1 kernels, A/B of the short-row kernels
set -u
O=gpurun_out/r02c1; mkdir -p $O
timeout 120 build/mfma4x4_probe > $O/probe.txt 2>&1
B="python bench.py --no-cpu-baseline"
timeout 300 $B --steps 10 --warmup 2 > $O/c3.json 2> $O/c3.err
IMP_SHORT_TEAM1=1 timeout 200 $B --no-topk --steps 10 --warmup 2 > $O/c3_short_team1.json 2> $O/c3_short_team1.err
IMP_CG_STATS=1 timeout 200 $B --no-topk --steps 1 --warmup 1 > $O/c3_stats.json 2> $O/c3_stats.err
timeout 300 $B --no-topk --shape c2 --factors 64 --solver cholesky --steps 3 --warmup 1 > $O/c2_chol.json 2> $O/c2_chol.err
timeout 300 $B --no-topk --shape c2 --factors 64 --solver cg --steps 3 --warmup 1 > $O/c2_cg.json 2> $O/c2_cg.err
timeout 300 $B --no-topk --shape ml20m --factors 256 --solver cg --steps 3 --warmup 1 > $O/c5_cg256.json 2> $O/c5_cg256.err
timeout 300 $B --no-topk --shape ml20m --factors 128 --solver cg --steps 3 --warmup 1 > $O/c5_cg128.json 2> $O/c5_cg128.err
ls -la $O
