#!/bin/bash
# round 4: Cholesky f = 64 with four columns per factorisation step: parity + A/B against the column-wise form
set -u
TAG=${1:-r5o}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py -x -q -m gpu -k "chol or golden" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -5
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky"
for rep in 1 2; do
timeout 600 $B > $O/chol_new$rep.json 2> $O/chol_new$rep.err
IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_cholcol.so timeout 600 $B > $O/chol_col$rep.json 2> $O/chol_col$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/chol_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[0]); print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], {k:round(v,2) for k,v in j["kernels_ms_per_step"].items()})
PY
IMP_CHOL_STATS=1 timeout 600 $B 2>&1 >/dev/null | grep chol-stats | head -4
