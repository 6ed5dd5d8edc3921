#!/bin/bash
# round 5: the split-bf16 gramian at f = 128 (IMP_GRAM_BF16X3=1, opt-in) against the fp32 matrix instruction: the headline step with its
# per-kernel table and the core clock behind a step, both forms alternating on one box, 20 and 300 timed steps
set -u
O=gpurun_out/r5g; mkdir -p $O
for steps in 20 300; do
for form in fp32 bf3 fp32 bf3; do
  if [ $form = bf3 ]; then export IMP_GRAM_BF16X3=1; else unset IMP_GRAM_BF16X3; fi
  python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-topk --no-extras > $O/bench_${form}_$steps.json 2> $O/bench_${form}_$steps.err
  python - <<PY
import json
d = json.loads(open("$O/bench_${form}_$steps.json").read().strip().splitlines()[-1])
k = d.get("kernels_ms_per_step", {})
print("$form", $steps, "ms_per_step %.4f" % d["ms_per_step"], "clock", d["core_clock_mhz"]["before_warmup"], d["core_clock_mhz"]["behind_a_step"],
      {a: round(b, 4) for a, b in k.items() if "gram" in a or "short" in a or "team2" in a})
PY
done
done
