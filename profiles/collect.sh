#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to.  Run on the GPU box from the repo root:
#   bash profiles/collect.sh r03      (writes gpurun_out/prof_r03/..., summaries are then copied into profiles/)
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa tracing), as the pool requires;
# FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) cannot share a pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).
# The SAME bench command also runs once WITHOUT the profiler on the same box (<tag>_bench_unprofiled.json): kernels run
# slower under rocprofv3 (lower sustained clock, MI355X_MICROARCH.md "DVFS give-back" item 2), so the profiled averages
# are compared with the profiled run's own HIP events (<tag>_bench_under_rocprof.json) and the un-profiled events are kept
# beside them.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-topk --no-extras"
ONLY=${2:-all}   # `collect.sh r03 topk`: only the top-k passes (summarize.py merges them into the tag's existing summaries)
if [ "$ONLY" = "all" ]; then
# (bench.py prints one SHORT line; the full record -- what summarize.py reads -- goes to the side file IMP_BENCH_DETAIL names)
IMP_BENCH_DETAIL="$O/bench_unprofiled.json" $B > "$O/bench_unprofiled.line" 2> "$O/unprofiled.err"
IMP_BENCH_DETAIL="$O/bench_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $B > "$O/bench_under_rocprof.line" 2> "$O/stats.err"
IMP_BENCH_DETAIL=/dev/null rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- $B > /dev/null 2>&1
IMP_BENCH_DETAIL=/dev/null rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/write" -- $B > /dev/null 2>&1
IMP_BENCH_DETAIL=/dev/null rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$O/sq" -- $B > /dev/null 2>&1
fi
# secondary paths: top-k (emit path) and the configs[1] Cholesky: kernel statistics + one counter pass each
T="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
C="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky"
IMP_BENCH_DETAIL="$O/topk_bench_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$O/topk_stats" -- $T > /dev/null 2> "$O/topk_stats.err"
IMP_BENCH_DETAIL=/dev/null rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/topk_pmc" -- $T > /dev/null 2>&1
if [ "$ONLY" = "all" ]; then
IMP_BENCH_DETAIL="$O/chol_bench_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$O/chol_stats" -- $C > /dev/null 2> "$O/chol_stats.err"
IMP_BENCH_DETAIL=/dev/null rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/chol_pmc" -- $C > /dev/null 2>&1
fi
python "$R/profiles/summarize.py" "$O" "$TAG"
cp "$R"/profiles/${TAG}_* "$O"/ 2>/dev/null   # only gpurun_out/ travels back from the GPU box
rm -rf "$O"/stats "$O"/fetch "$O"/write "$O"/sq "$O"/topk_stats "$O"/topk_pmc "$O"/chol_stats "$O"/chol_pmc
ls -la "$O"
