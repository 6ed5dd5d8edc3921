#!/bin/bash
# round 4: factor-grid parity tests + PARITY.md generation
set -u
TAG=${1:-r5e}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_als.py -x -q -m gpu -k "factor_grid or 1024 or other_factor" -rP > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -h "rel=\|passed\|failed\|rc=" $O/tests.log | tail -12
(time timeout 1200 python profiles/parity_report.py $O) > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/parity.log | tail -45
