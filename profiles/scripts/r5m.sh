#!/bin/bash
# full -m gpu suite + driver-style bench
set -u
TAG=${1:-r5m}; O=gpurun_out/$TAG; mkdir -p $O
(time timeout 2400 python -m pytest tests -x -q -m gpu) > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -12
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[0])
print("ms/step", j["ms_per_step"], "value", j["value"], "roofline", {k:j["roofline"][k] for k in ("frac","frac_half_sweep_events","avg_launch_ms")})
for k in ("fit_c3","fp16_c3","cholesky_c2","cg_c2","cg_c5","similar_items_c5","c4_full_1gpu","c4_shard","cg_c3_f32","cg_c3_f64","cg_c3_f192","cg_c3_f256"):
    v=j.get(k)
    if v: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_iter","compute_ms_per_iter","items_per_s","setup_s","fit_s","updates_per_s")}, (v.get("roofline") or {}).get("frac"))
print([k for k in j if k.endswith("_error")], j.get("extras_s"))
t=j["topk"]; print("topk", t["value"], t["model_recommend_recs_per_s"])
PY
tail -3 $O/bench.err
