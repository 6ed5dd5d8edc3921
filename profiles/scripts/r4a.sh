#!/bin/bash
# cluster kernels: nap between two polls of an exchange (configs[1]-shaped CG, where the clusters are a third of the time)
set -u
TAG=${1:-r4a}; O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-topk --no-extras --steps 4 --warmup 1 --shape c2 --factors 64 --solver cg"
timeout 300 $B > $O/b_nap2.json 2> $O/b_nap2.err
for v in cnap0 cnap1 cnap4; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 300 $B > $O/b_$v.json 2> $O/b_$v.err; done
python - $O <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    j = json.load(open(f)); print(os.path.basename(f), "ms/iter %.3f" % j["ms_per_step"], {k.replace("als_cg_", ""): round(v, 3) for k, v in j["kernels_ms_per_step"].items() if "cluster" in k})
PY
