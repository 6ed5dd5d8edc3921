import json, sys, glob, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "*.json"))):
    try:
        j = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = j.get("roofline") or {}
    print(os.path.basename(f), "ms/step %.3f" % j["ms_per_step"], "frac", round(r.get("frac", 0), 3), r.get("row_class"))
    print("   ", {k.replace("als_cg_", ""): round(v, 3) for k, v in j["kernels_ms_per_step"].items()})
    if j.get("row_classes"):
        print("   ", {k: (round(v["ms_per_step"], 3), round(v["achieved_GBps"])) for k, v in j["row_classes"].items()})
    for k in ("topk",):
        if k in j: print("   topk", round(j[k]["value"]), {a: round(b, 4) for a, b in j[k]["kernels_ms_per_batch"].items()})
for f in sorted(glob.glob(os.path.join(d, "*stats.err"))):
    seen = set()
    for line in open(f):
        if line.startswith("[cg-stats]"):
            key = line.split()[1] + line.split()[2]
            if key not in seen:
                seen.add(key); print(line.strip())
for f in sorted(glob.glob(os.path.join(d, "tests*.log"))):
    print(open(f).read()[-400:])
