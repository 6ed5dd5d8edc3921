#!/bin/bash
# round 4: where do the rounds of the normal-matrix kernel spend their cycles?  kernel statistics + two SQ counter passes
set -u
TAG=${1:-r6b}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-topk --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- $B > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- $B > /dev/null 2> $O/sq2.err
cd $R
python - <<PY
import csv, glob, collections
O="$O"
for f in glob.glob(O+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nm" in r["Name"] or "qfteam" in r["Name"] or "qfgroup" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
for d in ("sq1","sq2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "nm_kernel" in k or "qfteam" in k:
                k=k[:60]
                acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        # count dispatches
    for k,v in acc.items():
        print(d, k)
        for c,x in sorted(v.items()): print("   ", c, "%.4g" % x)
PY
rm -rf $O/stats $O/sq1 $O/sq2
