C="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-topk --no-extras --shape c2 --factors 64 --solver cholesky"
for rep in 1 2; do for v in base chw2; do
L=$PWD/implicit_amd/libimplicit_hip.so; [ $v = chw2 ] && L=$PWD/build/variants/libimplicit_hip_chw2.so
IMP_LIB_PATH=$L IMP_BENCH_DETAIL=/dev/null $C 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done; done
