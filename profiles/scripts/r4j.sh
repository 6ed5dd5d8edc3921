#!/bin/bash
# top-k: query rows split to bf16 terms once per call (emit path) + fp16 operands kept raw until the split: parity + A/B
set -u
O=gpurun_out/${1:-r4j}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_golden.py tests/test_gpu_model.py tests/test_gpu_round2.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout 300 $B > $O/c3.json 2> $O/c3.err
IMP_TOPK_NO_QSPLIT=1 timeout 300 $B > $O/c3_noqsplit.json 2> $O/c3_noqsplit.err
timeout 300 $B > $O/c3b.json 2> $O/c3b.err
python - <<PY
import json
for n in ("c3","c3_noqsplit","c3b"):
    d=json.load(open("$O/%s.json"%n))["topk"]
    print(n, round(d["value"]), round(d["scoring_TFLOPs"],1), {k:round(v,4) for k,v in d["kernels_ms_per_batch"].items()})
PY
# fp16 factors through the same call
timeout 300 python - <<'PY' > $O/fp16.txt 2>&1
import time, numpy as np
import implicit_amd.gpu as gpu
rng = np.random.default_rng(0)
Y = (rng.random((292385, 128), dtype=np.float32) - 0.5) * 0.2
Q = (rng.random((1000, 128), dtype=np.float32) - 0.5) * 0.2
for dt in (np.float32, np.float16):
    Yd, Qd = gpu.Matrix(Y.astype(dt)), gpu.Matrix(Q.astype(dt))
    knn = gpu.KnnQuery()
    knn.topk(Yd, Qd, 10)
    t = time.perf_counter()
    for _ in range(20): knn.topk(Yd, Qd, 10)
    dtm = (time.perf_counter() - t) / 20
    print(np.dtype(dt).name, "ms per 1000-query call %.3f" % (dtm * 1e3))
PY
cat $O/fp16.txt
