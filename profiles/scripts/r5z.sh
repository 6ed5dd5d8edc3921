#!/bin/bash
# round 5, verification run: the whole -m gpu suite, the driver-style bench line, one collect.sh r05 run (kernel stats, PMC passes,
# roofline check), PARITY.md regenerated.  Everything lands under gpurun_out/r5z (the r05_* summaries are copied to profiles/ by hand).
set -u
O=gpurun_out/r5z; mkdir -p $O
python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $O/pytest.log
tail -3 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "events frac", d["roofline"]["frac_half_sweep_events"])
for k in ("cg_c3_f192", "cg_c3_f256", "cg_c5", "cg_c2", "cholesky_c2", "fp16_c3", "c4_full_1gpu", "c4_shard"):
    if k in d: print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items() if a in ("ms_per_iter", "compute_ms_per_iter")}, round(d[k].get("roofline", {}).get("frac", 0), 3))
print("topk", d.get("topk", {}).get("value"), d.get("topk", {}).get("model_recommend_recs_per_s"))
PY
bash profiles/collect.sh r05 > $O/collect.log 2>&1
tail -3 $O/collect.log
python profiles/parity_report.py $O/parity > $O/parity.log 2>&1
tail -2 $O/parity.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
