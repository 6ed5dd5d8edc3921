#!/bin/bash
# fit() set-up with the transpose on a worker thread beside the user-side upload and the initial factors: parity + wall clock
set -u
O=gpurun_out/${1:-r4t}; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 200 python - > $O/fit_setup.txt 2>&1 <<'PY'
import sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
from implicit_amd.als import AlternatingLeastSquares
C = named("lastfm360k")
for rep in range(4):
    m = AlternatingLeastSquares(factors=128, iterations=2, random_state=1, use_gpu=True)
    times = []
    t0 = time.perf_counter(); m.fit(C, show_progress=False, callback=lambda it, dt, loss: times.append(dt)); tot = time.perf_counter() - t0
    print(f"fit(2 iterations) {tot:.3f} s, iterations {sum(times):.3f} s, set-up {tot - sum(times):.3f} s")
PY
cat $O/fit_setup.txt
