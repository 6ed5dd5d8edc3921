#!/bin/bash
# round 5: the fp16 form of the emit GEMM (topk.hip H2) against the six-product bf16 form (IMP_TOPK_BF16X3=1): top-k tests, then the
# bench's top-k object with per-kernel times, both forms on the same box
set -u
O=gpurun_out/r5t; mkdir -p $O
python -m pytest tests/test_gpu_topk.py tests/test_gpu_round2.py -x -q -m gpu -k "topk or emit or fp16_form or TOPK or knn" 2>&1 | tail -4 > $O/pytest.log
tail -3 $O/pytest.log
for form in ${FORMS:-h2 bf16x3 h2 bf16x3}; do
  if [ $form = bf16x3 ]; then export IMP_TOPK_BF16X3=1; else unset IMP_TOPK_BF16X3; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_$form.json 2> $O/bench_$form.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$form.json").read().strip().splitlines()[-1])["topk"]
print("$form", round(d["value"]), round(d.get("model_recommend_recs_per_s", 0)), {k: round(v, 4) for k, v in d["kernels_ms_per_batch"].items()})
PY
done
