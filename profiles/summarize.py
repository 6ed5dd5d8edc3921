#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by profiles/collect.sh into two small tracked files:
profiles/<tag>_kernel_stats.csv (the --stats table) and profiles/<tag>_pmc_summary.json (per-kernel counter
averages per dispatch + corrected HBM traffic).  Units/corrections follow MI355X_MICROARCH.md section HBM:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("imp::", "")


stats = glob.glob(os.path.join(out_dir, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(here, f"{tag}_kernel_stats.csv"))
bench = os.path.join(out_dir, "bench_under_rocprof.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(here, f"{tag}_bench_under_rocprof.json"))

summary = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(out_dir, sub, "*", "*counter_collection.csv")):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            agg[key] += float(row["Counter_Value"])
            cnt[key] += 1
        for (k, c), v in agg.items():
            summary[k][c] = v / cnt[(k, c)]
            summary[k]["dispatches_" + sub] = cnt[(k, c)]
for k, d in summary.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_per_dispatch_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes_per_dispatch"] = d["WRITE_SIZE"] * 1024.0
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
json.dump(summary, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(f"wrote profiles/{tag}_kernel_stats.csv and profiles/{tag}_pmc_summary.json ({len(summary)} kernels)")
