#!/bin/bash
set -u
O=gpurun_out/${1:-chol}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_als.py tests/test_gpu_golden.py tests/test_gpu_model.py tests/test_gpu_round2.py -x -q -m gpu -k "chol or golden or checkerboard" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky --steps 3 --warmup 1 > $O/c2_chol.json 2> $O/c2_chol.err
python -c "
import json;d=json.load(open('$O/c2_chol.json'));print('c2 cholesky ms/iter', d['ms_per_step'], d['kernels_ms_per_step'])"
