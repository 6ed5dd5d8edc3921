#!/bin/bash
# round 4: logical-shard harness on the GPU + wave -> SIMD placement probe
set -u
TAG=${1:-r5c}; O=gpurun_out/$TAG; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/wave_simd_map profiles/micro/wave_simd_map.hip > /dev/null 2>&1 && /tmp/wave_simd_map > $O/wave_simd_map.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_logical_shards.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/tests.log | tail -25; grep "block" $O/wave_simd_map.txt | head -20
