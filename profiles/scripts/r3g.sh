#!/bin/bash
# round 3 checkpoint: the whole -m gpu suite, the driver-style default bench (with extras; wall time recorded), fit() set-up breakdown
set -u
TAG=${1:-r3g}; O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? wall $(( $(date +%s) - t0 ))s" >> $O/tests.log
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 ))s" >> $O/bench.err
python - > $O/fit_setup.txt 2>&1 <<'PY'
import sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
from implicit_amd.utils import check_random_state, random_factors, transpose_csr
C = named("lastfm360k")
def t(label, fn):
    t0 = time.perf_counter(); r = fn(); gpu.synchronize(); print(f"{label:40s} {1e3*(time.perf_counter()-t0):8.1f} ms"); return r
for rep in range(2):
    print("rep", rep)
    Ciu = t("transpose_csr (library, threaded)", lambda: transpose_csr(C))
    t("C.T.tocsr() (scipy)", lambda: C.T.tocsr())
    rng = check_random_state(1)
    x0 = t("random_factors users (threaded)", lambda: random_factors(rng, C.shape[0], 128))
    t("rng.random users (plain)", lambda: rng.random((C.shape[0], 128), dtype=np.float32) * 0.01)
from implicit_amd.als import AlternatingLeastSquares
for rep in range(2):
    m = AlternatingLeastSquares(factors=128, iterations=2, random_state=1, use_gpu=True)
    times = []
    t0 = time.perf_counter(); m.fit(C, show_progress=False, callback=lambda it, dt, loss: times.append(dt)); tot = time.perf_counter() - t0
    print(f"fit(2 iterations) {tot:.3f} s, iterations {sum(times):.3f} s, set-up {tot - sum(times):.3f} s")
PY
tail -3 $O/tests.log; tail -2 $O/bench.err; cat $O/fit_setup.txt
python - $O <<'PY'
import json
j = json.load(open(__import__("sys").argv[1] + "/bench.json"))
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"])
for k in ("fit_c3", "fp16_c3", "cholesky_c2", "cg_c2", "cg_c5", "similar_items_c5", "c4_full_1gpu", "c4_shard", "topk"):
    v = j.get(k)
    if isinstance(v, dict):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list, str))}, (v.get("roofline") or {}).get("frac"))
print({k: v for k, v in j.items() if k.endswith("_error")}, j.get("extras_s"))
PY
