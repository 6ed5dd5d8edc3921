#!/bin/bash
# round 4: top-k device outputs, sharded recommend, model.recommend overhead
set -u
TAG=${1:-r5f}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_model.py tests/test_gpu_logical_shards.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -15
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.load(open("$O/bench.json"))
t=j["topk"]; print("ms/step", j["ms_per_step"], "roofline", {k:j["roofline"][k] for k in ("frac","frac_half_sweep_events","avg_launch_ms","traffic")})
print("topk", t["value"], "model_recommend", t["model_recommend_recs_per_s"], t["roofline"]["frac"], t["roofline"]["traffic"])
print({k:round(v,4) for k,v in t["kernels_ms_per_batch"].items()})
PY
