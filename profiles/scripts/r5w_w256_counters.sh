#!/bin/bash
# round 5: rocprofv3 evidence for the f = 256 resident kernels (als_cg_w256.hip) on the configs[2] matrix at f = 256: kernel statistics,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes, FETCH doubled for gfx950 as MI355X_MICROARCH.md prescribes) and two SQ
# passes (vector / LDS / matrix-pipe activity).  Writes gpurun_out/r5w/w256_counters.txt.
set -u
TAG=${1:-r5w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/profiles/scripts/r5c_f256.py c3_256"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/run_stats.log 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- $B > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- $B > /dev/null 2> $O/sq2.err
cd $R
export W256_OUT=$O
python - > $O/w256_counters.txt <<'PY'
import csv, glob, collections
import os
O=os.environ["W256_OUT"]
print("# rocprofv3 on profiles/scripts/r5c_f256.py c3_256 (configs[2] matrix, f = 256, CG 3; 1 warm-up + 3 timed iterations = 8 launches per kernel)")
print("## kernel statistics")
for f in glob.glob(O+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "w256" in r["Name"] or "f256" in r["Name"] or "cg_long" in r["Name"]:
            print("%-60s calls %4s  avg %10.1f us" % (r["Name"].replace("void imp::(anonymous namespace)::","").replace("void imp::","")[:60], r["Calls"], float(r["AverageNs"])/1e3))
print("## counters per dispatch (averages); FETCH_SIZE / WRITE_SIZE in KiB, corrected HBM read bytes = 2 x FETCH_SIZE x 1024")
for d in ("fetch","write","sq1","sq2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.Counter())
    for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "w256_kernel" in k:
                k=k.replace("void imp::(anonymous namespace)::","")[:40]
                acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
    for k,v in sorted(acc.items()):
        print(d, k)
        for c,x in sorted(v.items()): print("    %-28s %.5g" % (c, x / max(1, n[k][c])))
PY
cat $O/w256_counters.txt | head -80
rm -rf $O/stats $O/fetch $O/write $O/sq1 $O/sq2
