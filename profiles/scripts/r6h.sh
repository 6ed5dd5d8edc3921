#!/bin/bash
# round 4, after the normal-matrix kernels: full -m gpu suite, PARITY.md, driver-style bench, smoke
set -u
TAG=${1:-r6h}; O=gpurun_out/$TAG; mkdir -p $O
(time timeout 2400 python -m pytest tests -q -m gpu) > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -8
timeout 600 python profiles/parity_report.py $O > $O/parity.log 2>&1; echo "parity rc=$?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[0])
print("ms/step", j["ms_per_step"], "value", j["value"], "roofline", {k:j["roofline"][k] for k in ("frac","frac_half_sweep_events","avg_launch_ms","traffic")})
for k in ("fit_c3","fp16_c3","cholesky_c2","cholesky_c3_f128","cg_c2","cg_c5","similar_items_c5","c4_full_1gpu","c4_shard","cg_c3_f32","cg_c3_f64","cg_c3_f192","cg_c3_f256"):
    v=j.get(k)
    if v: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_iter","compute_ms_per_iter","items_per_s","setup_s","fit_s")}, (v.get("roofline") or {}).get("frac"))
print([k for k in j if k.endswith("_error")], j.get("extras_s"))
t=j["topk"]; print("topk", t["value"], t["model_recommend_recs_per_s"], t["roofline"]["traffic"])
PY
tail -3 $O/bench.err
