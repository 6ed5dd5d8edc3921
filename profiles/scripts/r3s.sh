#!/bin/bash
# generic-f CG path at the C3 shape (f = 100 / 32 / 16 / 96) against the resident kernels (f = 128 / 64); forced-sharded one-rank run
set -u
TAG=${1:-r3s}; O=gpurun_out/$TAG; mkdir -p $O
for f in 128 100 96 64 50 32 16; do
  timeout 300 python bench.py --no-cpu-baseline --no-topk --no-extras --steps 4 --warmup 1 --factors $f > $O/f$f.json 2> $O/f$f.err
done
IMP_FORCE_SHARDED=1 timeout 600 python bench.py --gpus 1 --shape c4 --scale 0.05 --steps 3 --warmup 1 > $O/forced_sharded.json 2> $O/forced_sharded.err; echo "forced rc=$?" >> $O/forced_sharded.err
python - $O <<'PY'
import json, glob, os, sys
for f in (128, 100, 96, 64, 50, 32, 16):
    try:
        j = json.load(open(sys.argv[1] + f"/f{f}.json")); print("f", f, "ms/iter %.3f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernels_ms_per_step"].items()})
    except Exception as e: print(f, e)
j = json.load(open(sys.argv[1] + "/forced_sharded.json")); print("forced sharded", j["ms_per_step"], j["config"]["parallelism"])
PY
tail -2 $O/forced_sharded.err
