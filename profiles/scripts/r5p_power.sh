#!/bin/bash
# round 5: clocks and power while the headline step runs, fp32 gramian (default) against the split-bf16 one (IMP_GRAM_BF16X3=1)
set -u
O=gpurun_out/r5p; mkdir -p $O
for form in fp32 bf3; do
  if [ $form = bf3 ]; then export IMP_GRAM_BF16X3=1; else unset IMP_GRAM_BF16X3; fi
  python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-topk --no-extras > $O/bench_$form.json 2> $O/bench_$form.err &
  BP=$!
  sleep 14   # import, matrix, plans
  for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.4; done > $O/smi_$form.txt
  wait $BP
  python - <<PY
import json
d = json.loads(open("$O/bench_$form.json").read().strip().splitlines()[-1])
print("$form", "ms_per_step %.4f" % d["ms_per_step"])
PY
  tail -4 $O/smi_$form.txt | cut -c1-400
done
