#!/bin/bash
# pmc_ab.sh <tag>: SQ / GRBM / LDS counters of the CG kernels, round-2 team kernels (IMP_TEAM_FUSED=0) against the current
# ones, same box.  Counter passes carry --kernel-trace only.  Output: gpurun_out/<tag>/pmc_{old,new}.json
set -u
TAG=${1:-pmcab}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-topk --no-extras"
for arm in old new; do
  if [ $arm = old ]; then export IMP_TEAM_FUSED=0; else unset IMP_TEAM_FUSED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${arm}_stats -- $B > $O/${arm}_bench.json 2> $O/${arm}_stats.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${arm}_p1 -- $B > /dev/null 2> $O/${arm}_p1.err
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${arm}_p2 -- $B > /dev/null 2> $O/${arm}_p2.err
  python - "$O" $arm <<'PY'
import collections, csv, glob, json, os, sys
O, arm = sys.argv[1], sys.argv[2]
summary = collections.defaultdict(dict)
for sub in ("p1", "p2"):
    for f in glob.glob(os.path.join(O, f"{arm}_{sub}", "*", "*counter_collection.csv")):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("imp::", "")
            key = (name, row["Counter_Name"])
            agg[key] += float(row["Counter_Value"]); cnt[key] += 1
        for (k, c), v in agg.items():
            summary[k][c] = v / cnt[(k, c)]
            summary[k]["dispatches"] = cnt[(k, c)]
for f in glob.glob(os.path.join(O, f"{arm}_stats", "*", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Name"].split("(")[0].replace("void ", "").replace("imp::", "")
        summary[name]["avg_ns"] = float(row["AverageNs"]); summary[name]["calls"] = int(row["Calls"])
json.dump(summary, open(os.path.join(O, f"pmc_{arm}.json"), "w"), indent=1, sort_keys=True)
for k, d in sorted(summary.items()):
    if "als_cg" in k and "avg_ns" in d and "GRBM_GUI_ACTIVE" in d:
        print(arm, k[:70], "avg_us %.1f" % (d["avg_ns"] / 1e3), "clock_GHz %.2f" % (d["GRBM_GUI_ACTIVE"] / d["avg_ns"]),
              "valu %.3g lds %.3g" % (d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_LDS", 0)),
              "lds_idx_active %.3g conflict %.3g" % (d.get("SQ_LDS_IDX_ACTIVE", 0), d.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
done
rm -rf $O/*_stats $O/*_p1 $O/*_p2
