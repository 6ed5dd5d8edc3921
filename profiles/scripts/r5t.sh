#!/bin/bash
# round 4: normal-matrix kernels for the long rows (als_cg_nm.hip): parity first, then the bench line
set -u
TAG=${1:-r5t}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nm.py -x -q -m gpu -s > $O/nm_tests.log 2>&1; echo "nm tests rc=$?" >> $O/nm_tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/nm_tests.log | tail -25
timeout 1200 python -m pytest tests/test_gpu_als.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -8
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[0])
print("ms/step", j["ms_per_step"], "value", j["value"], "roofline", {k:j["roofline"][k] for k in ("frac","frac_half_sweep_events","avg_launch_ms")})
print({k:(round(v["ms_per_step"],3), round(v["frac"],3)) for k,v in j["row_classes"].items()})
for k in ("fit_c3","fp16_c3","cg_c2","cg_c5","c4_full_1gpu","c4_shard","cg_c3_f64","cg_c3_f192"):
    v=j.get(k)
    if v: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_iter","compute_ms_per_iter")}, (v.get("roofline") or {}).get("frac"), {a:round(b,2) for a,b in (v.get("kernels_ms_per_iter") or {}).items() if "nm" in a or "long" in a or "cluster" in a})
print([k for k in j if k.endswith("_error")], j.get("extras_s"))
PY
tail -3 $O/bench.err
