#!/bin/bash
# zero-padded route for factor counts other than 64 / 128: parity + speed at the C3 shape
set -u
TAG=${1:-r3w}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for f in 100 50 16; do
  timeout 300 python bench.py --no-cpu-baseline --no-topk --no-extras --steps 4 --warmup 1 --factors $f > $O/f$f.json 2> $O/f$f.err
done
python - $O <<'PY'
import json, sys
for f in (100, 50, 16):
    j = json.load(open(sys.argv[1] + f"/f{f}.json")); print("f", f, "ms/iter %.3f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernels_ms_per_step"].items() if "pad" in k or "team2" in k or "short" in k})
PY
grep -E "padded route|f=100|f=50 " $O/tests.log | head; tail -3 $O/tests.log
