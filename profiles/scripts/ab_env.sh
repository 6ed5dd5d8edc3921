#!/bin/bash
# A/B of environment switches on the headline bench: ab_env.sh <tag> "VAR=val" "VAR2=val" ...  ("" = default)
out=gpurun_out/${1:-ab}; shift
mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-topk ${BENCH_ARGS:---no-extras} > $out/b$i.json 2> $out/b$i.err
  echo "== $setting" >> $out/summary.txt
  python profiles/scripts/show.py $out 2>/dev/null | grep -A2 "b$i.json" >> $out/summary.txt
done
cat $out/summary.txt
