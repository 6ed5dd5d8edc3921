for k in 1 2; do IMP_LIB_PATH=$PWD/build/variants/libimplicit_hip_scrko$k.so IMP_BENCH_DETAIL=gpurun_out/scrko$k.json python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras >/dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/scrko$k.json'))['topk']['kernels_ms_per_batch']; print('ko$k', round(d['topk_select_candidates'],4))"; done
