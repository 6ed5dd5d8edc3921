#!/bin/bash
# knock-outs of the emit GEMM (timing only; every variant also drops the emit epilogue so that garbage scores cannot flood it)
set -u
O=gpurun_out/${1:-r4n}; mkdir -p $O
timeout 200 python profiles/scripts/topk_kernels.py > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log
for v in ko_epi ko_epi_split ko_epi_mfma ko_epi_dma ko_epi_qload ko_epi_loads ko_epi_loads_split; do
  IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_$v.so timeout 200 python profiles/scripts/topk_kernels.py > $O/$v.log 2>&1; echo "rc=$?" >> $O/$v.log
done
for f in $O/*.log; do echo "== $f"; tail -4 $f | cut -c1-400; done
