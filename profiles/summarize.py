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
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("imp::", "")


stats = glob.glob(os.path.join(out_dir, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(here, f"{tag}_kernel_stats.csv"))
for name in ("bench_under_rocprof.json", "bench_unprofiled.json", "topk_bench_under_rocprof.json", "chol_bench_under_rocprof.json"):
    if os.path.exists(os.path.join(out_dir, name)):
        shutil.copy(os.path.join(out_dir, name), os.path.join(here, f"{tag}_{name}"))
for sub in ("topk", "chol"):
    f = glob.glob(os.path.join(out_dir, f"{sub}_stats", "*", "*kernel_stats.csv"))
    if f:
        shutil.copy(f[0], os.path.join(here, f"{tag}_{sub}_kernel_stats.csv"))


def per_side(trace_csv):
    """The CG kernels run once per HALF SWEEP: the same instantiation is dispatched for the user side (even occurrences) and
    for the item side (odd ones), on different row sets -- the min / max columns of the --stats table are those two, not
    jitter.  Averages per side from the kernel trace."""
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    seen, sides = collections.Counter(), collections.defaultdict(lambda: [[], []])
    for r in rows:
        k = short(r["Kernel_Name"])
        sides[k][seen[k] % 2].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        seen[k] += 1
    out = {}
    for k, (a, b) in sides.items():
        if "als_cg" in k and "long" not in k and "reset" not in k and len(a) == len(b) and a:
            out[k] = {"user_side_avg_us": sum(a) / len(a), "item_side_avg_us": sum(b) / len(b), "launches_per_side": len(a)}
    return out


def whole_step_check(out_dir, stats_csv):
    """bench.py's `roofline` is the WHOLE step (round 4): algorithmic bytes per step / ms_per_step.  The same fraction three
    ways: from rocprofv3's per-kernel totals of the profiled run (every CG kernel, / the steps it executed: timed + warm-up +
    the 3 detail iterations), from that run's own HIP events around the half sweeps, and from the un-profiled run."""
    rec = {}
    for label in ("bench_under_rocprof.json", "bench_unprofiled.json"):
        path = os.path.join(out_dir, label)
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                rl = j["roofline"]
                rec[label] = {"ms_per_step": j["ms_per_step"], "frac": rl["frac"], "half_sweep_ms_hip_events": rl["avg_launch_ms"],
                              "frac_half_sweep_events": rl["frac_half_sweep_events"],
                              "algorithmic_bytes_per_step": rl["algorithmic_bytes_per_step"], "steps": j["steps"], "warmup": j["warmup"],
                              # (round 5: three more steps, each with the core-clock probe queued behind it)
                              "probe_steps": len(j.get("core_clock_mhz", {}).get("behind_a_step", []))}
            except Exception as e:  # noqa: BLE001
                rec[label] = {"error": str(e)}
    prof = rec.get("bench_under_rocprof.json", {})
    if os.path.exists(stats_csv) and "steps" in prof:
        steps_run = prof["steps"] + prof["warmup"] + min(prof["steps"], 3) + prof.get("probe_steps", 0)
        total_ns = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(stats_csv))
                       if "als_cg" in r["Name"] or "cg_long" in r["Name"])
        ms = total_ns / 1e6 / steps_run
        rec["whole_step_from_rocprof"] = {"cg_kernel_ms_per_step": ms, "steps_in_the_profile": steps_run,
                                          "frac_of_8TBps": prof["algorithmic_bytes_per_step"] / (ms * 1e-3) / 1e9 / 8000.0,
                                          "note": "sum of rocprofv3 TotalDurationNs over every CG kernel / steps executed; the HIP-event "
                                                  "figure of the same run is 2 x half_sweep_ms_hip_events (launch gaps inside a half sweep included)"}
    return rec


traces = glob.glob(os.path.join(out_dir, "stats", "*", "*kernel_trace.csv"))
if traces:
    sides = per_side(traces[0])
    rec = {"per_side": sides}
    rec.update(whole_step_check(out_dir, os.path.join(here, f"{tag}_kernel_stats.csv")))
    json.dump(rec, open(os.path.join(here, f"{tag}_roofline_check.json"), "w"), indent=1, sort_keys=True)

summary = collections.defaultdict(dict)
prev = os.path.join(here, f"{tag}_pmc_summary.json")
if os.path.exists(prev):  # a partial collection (collect.sh <tag> topk) replaces only the kernels it saw
    for k, d in json.load(open(prev)).items():
        summary[k] = d
seen_now = set()
for sub in ("fetch", "write", "sq", "topk_pmc", "chol_pmc"):  # what a pass collects again it replaces wholesale (kernel names change)
    if glob.glob(os.path.join(out_dir, sub, "*", "*counter_collection.csv")):
        for k in [k for k, d in summary.items() if "dispatches_" + sub in d]:
            del summary[k]
for sub in ("fetch", "write", "sq", "topk_pmc", "chol_pmc"):
    for f in glob.glob(os.path.join(out_dir, sub, "*", "*counter_collection.csv")):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            agg[key] += float(row["Counter_Value"])
            cnt[key] += 1
        for (k, c), v in agg.items():
            if k not in seen_now:
                seen_now.add(k)
                summary[k] = {}
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
