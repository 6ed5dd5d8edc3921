#!/usr/bin/env python3
"""bench.py -- ALS training throughput on MI355X (BASELINE.json metric: "ALS user+item updates/sec
per iteration (factors=128); top-k recs/sec").

A *step* is one full ALS iteration of the hot path over the synthetic confidence matrix: user half
sweep (gramian YtY + per-row 3-step CG solves) then item half sweep (gramian XtX + solves), plus --
for N > 1 -- the RCCL exchange of the gramians and of the freshly solved factor shards.  Inputs are
resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 workload: BASELINE configs[2] -- last.fm-360K-shaped synthetic CSR (358,868 users x 292,385
items, ~17.5M nnz requested), factors=128, CG cg_steps=3, fp32.  N > 1: STRONG scaling on BASELINE
configs[3] (10M users x 1M items x 500M nnz, f=128; the matrix is a fixed 8 x 8 grid of blocks, so
N = 1, 2, 4, 8 factorise the same matrix; users and items row-sharded over the ranks).  The one-GPU
point of that curve is `--gpus 1 --shape c4` (the same sharded driver with one rank); the default
N = 1 line also carries it as the extra `c4_full_1gpu`.  `--weak`: one configs[2]-shaped shard per rank.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the library
stream inside the timed region) and `cpu_baseline` (the reference's own Cython CPU solver from
oracle/_ref, or the plain-C oracle port, on a bounded row sample of the same workload).
"""
import argparse
import json
import os
import sys
import time
import warnings

# scipy-openblas is built for <= 64/128 threads and aborts on the 256-core GPU box; the reference wants
# single-threaded BLAS under its OpenMP loop anyway (implicit/utils.py:18-62)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FACTORS = 128
REG = 0.01
CG_STEPS = int(os.environ.get("IMP_BENCH_CG_STEPS", "3"))  # debug override; the metric is quoted at 3
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="lastfm360k")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; flagged in the output)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-topk", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--solver", choices=["cg", "cholesky"], default="cg", help="cholesky = BASELINE configs[1] style run")
    ap.add_argument("--factors", type=int, default=FACTORS)
    ap.add_argument("--weak", action="store_true",
                    help="--gpus N > 1: weak scaling on one configs[2]-shaped shard per GPU instead of strong scaling on configs[3]")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (cholesky_c2, cg_c5, ...)")
    return ap.parse_args()


def cg_algorithmic_bytes(lengths, f):
    """nnz*(4f+8) + R*(8f+8) + 4f^2 for the rows one launch processes."""
    nnz = int(lengths.sum())
    rows = int(len(lengths))
    return nnz * (4 * f + 8) + rows * (8 * f + 8) + 4 * f * f


SHORT_ROW, LONG_ROW = 32, 512  # imp_csr::kShortRow / kLongRow
# schedule class -> the kernels that execute it (long rows: the normal-matrix kernels; with IMP_NM=0 one partial + one combine launch
# per CG pass)
CLASS_KERNELS = {"short": ["als_cg_short_rows"],
                 "mid": ["als_cg_team2_rows", "als_cg_team4_rows", "als_cg_team8_rows", "als_cg_team16_rows"],
                 "long": ["als_cg_long_partial", "als_cg_long_combine", "als_cg_nm_rows", "als_cg_nm_finish", "als_cg_fixup"]}


def class_bytes_per_iteration(Cui, Ciu, f):
    """Algorithmic bytes (single-pass traffic) of each row-length class over one iteration."""
    out = {k: 0 for k in CLASS_KERNELS}
    for M in (Cui, Ciu):
        lens = np.diff(M.indptr)
        for name, sel in (("short", (lens > 0) & (lens <= SHORT_ROW)), ("mid", (lens > SHORT_ROW) & (lens <= LONG_ROW)),
                          ("long", lens > LONG_ROW)):
            if sel.any():
                out[name] += cg_algorithmic_bytes(lens[sel], f)
    return out


# substring of the kernel function name -> (schedule class, dispatches per half sweep as a function of cg_steps)
PMC_KERNELS = {"als_cg_qfgroup_kernel": ("short", lambda s: 1), "als_cg_qfteam_kernel": ("mid", lambda s: 1),
               "cg_long_partial": ("long", lambda s: 1 + s), "cg_long_combine_kernel": ("long", lambda s: 1 + s),
               "als_cg_nm_kernel": ("long", lambda s: 1), "als_cg_nm_finish_kernel": ("long", lambda s: 1),
               "als_cg_nm_reduce_kernel": ("long", lambda s: 1), "als_cg_fault_fixup_kernel": ("long", lambda s: 1)}


def pmc_traffic_per_half_sweep(cg_steps):
    """HBM bytes per half sweep of each row class from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json, produced by profiles/collect.sh on this same bench command): corrected
    FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, summed over the class's kernels
    (the mid class runs four team widths = four kernel instantiations, each dispatched once per half sweep)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    summary = json.load(open(files[-1]))
    out = {}
    for kname, d in summary.items():
        for sub, (cls, per_sweep) in PMC_KERNELS.items():
            if sub in kname and "hbm_read_bytes_per_dispatch_corrected" in d:
                # cg_long_partial has two instantiations (first pass / later passes): 1 and cg_steps dispatches
                n = per_sweep(cg_steps)
                if sub == "cg_long_partial":
                    n = 1 if kname.rstrip(">").endswith("true") else cg_steps
                out[cls] = out.get(cls, 0.0) + n * (d["hbm_read_bytes_per_dispatch_corrected"] +
                                                     d.get("hbm_write_bytes_per_dispatch", 0.0))
    return out, os.path.basename(files[-1])


def _pmc_kernel_traffic(kernel_substr):
    """HBM read bytes per dispatch of one kernel from the newest committed PMC summary (None if it is not there)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None
    for kname, d in json.load(open(files[-1])).items():
        if kernel_substr in kname and "hbm_read_bytes_per_dispatch_corrected" in d:
            return d["hbm_read_bytes_per_dispatch_corrected"] + d.get("hbm_write_bytes_per_dispatch", 0.0)
    return None


def cpu_baseline(Cui, Ciu, X0, Y0, seconds):
    """Times the reference CPU solver (oracle/_ref, else the plain-C port) on a bounded row sample."""
    from oracle import oracle as port
    from oracle import ref

    als_ref, _ = ref.load()
    # OpenBLAS (reached by the reference from every OpenMP thread) supports at most 128 concurrent callers, and the
    # reference's dynamic OpenMP schedule does not scale across sockets: the thread count is probed (below) and the
    # fastest is used -- `cores` reports the threads actually used
    max_threads = min(os.cpu_count() or 1, 64)
    cores = max_threads
    kind = "reference" if als_ref is not None else "port"
    limiter = None
    if als_ref is not None:
        try:
            from threadpoolctl import threadpool_limits

            limiter = threadpool_limits(1, "blas")  # the reference demands single-threaded BLAS (utils.py:18-62)
        except ImportError:
            pass
    else:
        port.build()

    def run(M, A, B):
        A = A.copy()
        t = time.time()
        if als_ref is not None:
            als_ref.least_squares_cg(M, A, B, REG, num_threads=cores, cg_steps=CG_STEPS)
        else:
            port.least_squares_cg(M, A, B, REG, num_threads=cores, cg_steps=CG_STEPS)
        return time.time() - t

    # uniform row samples (every stride-th row: ids are popularity-ordered, a prefix would not be representative);
    # probe on ~2000 rows of each side, then size the sample for ~`seconds`
    def sample(M, A, n):
        stride = max(1, M.shape[0] // max(1, n))
        rows = np.arange(0, M.shape[0], stride)
        return M[rows], np.ascontiguousarray(A[rows])

    def timed(nu, ni):
        Mu, Au = sample(Cui, X0, nu)
        Mi, Ai = sample(Ciu, Y0, ni)
        return run(Mu, Au, Y0) + run(Mi, Ai, X0), Mu.shape[0], Mi.shape[0], int(Mu.nnz + Mi.nnz)

    best = None
    for cand in sorted({c for c in (8, 16, 32, max_threads) if c <= max_threads}):
        cores = cand
        t_c, pu, pi, _ = timed(4000, 4000)
        if best is None or t_c < best[0]:
            best = (t_c, cand, pu, pi)
    t_probe, cores, pu, pi = best
    rate = (pu + pi) / max(t_probe, 1e-6)
    frac = min(1.0, seconds * rate / (Cui.shape[0] + Ciu.shape[0]))
    t, su, si, nnz = timed(max(2000, int(Cui.shape[0] * frac)), max(2000, int(Ciu.shape[0] * frac)))
    if limiter is not None:
        limiter.restore_original_limits()
    return {
        "value": (su + si) / t,
        "unit": "updates/s",
        "cores": cores,
        "kind": kind,
        "sample_short": f"CG(3) f={X0.shape[1]} half sweeps, {su}+{si} rows, {t:.1f}s, {cores} threads",
        "sample": f"one CG(cg_steps=3,f={X0.shape[1]}) half-sweep over {su} users + {si} items taken at a uniform stride "
                  f"({nnz} nnz) of the same matrix, {t:.1f}s, OpenMP num_threads={cores} of {os.cpu_count()} logical cores, "
                  f"BLAS threads=1",
        "nnz_visits_per_s": nnz / t,
    }


def protect_stdout():
    """The contract is ONE JSON line on stdout.  librccl prints a version banner to the C-level stdout, which is buffered
    and surfaces when the process exits -- after the JSON line.  So fd 1 is pointed at stderr for the lifetime of the
    process (everything any library prints lands there) and the JSON line is written to a private duplicate of the
    original stdout."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def emit(saved_stdout, obj):
    os.write(saved_stdout, (json.dumps(obj) + "\n").encode())


def main():
    saved_stdout = protect_stdout()
    global FACTORS
    args = parse_args()
    FACTORS = args.factors
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N (N>1) must be launched with torch.distributed.run, one rank per GPU")
        args.gpus = world

    if os.environ.get("IMP_LIB_PATH"):  # tooling only: A/B of compile-time variants (python -m implicit_amd._build --variant ...)
        import implicit_amd._libpath as _libpath

        _libpath.OVERRIDE = os.environ["IMP_LIB_PATH"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import implicit_amd.gpu as gpu
    if not gpu.HAS_CUDA:
        sys.exit("bench.py: libimplicit_hip.so / HIP device unavailable (no CPU fallback exists)")
    from implicit_amd.synthetic import SHAPES, synthetic_csr

    gpu.set_device(local_rank)
    users, items, nnz_target, gamma = SHAPES[args.shape]
    if args.scale != 1.0:
        users, items, nnz_target = int(users * args.scale), int(items * args.scale), int(nnz_target * args.scale)

    # N > 1, or the one-GPU point of the configs[3] strong-scaling curve (`--gpus 1 --shape c4`); the env knob exercises
    # the N > 1 code path (chunked, pipelined exchange) with one rank
    if world > 1 or args.shape == "c4" or os.environ.get("IMP_FORCE_SHARDED"):
        from implicit_amd.gpu import sharded

        def rank0_roofline(Cui, Ciu, timed, steps):
            """Whole-step definition, as in the single-GPU line, on rank 0's shard: algorithmic bytes of every row class of
            both half sweeps / HIP-event time of its least_squares calls (one event pair per call, K chunk calls per half
            sweep) in the timed region.  The exchange is not in it (it runs on the other stream, overlapped)."""
            ms, n = timed.get("als_cg_half_sweep", (0.0, 0))
            if not n:
                return None
            per_step = sum(class_bytes_per_iteration(Cui, Ciu, FACTORS).values())
            achieved = per_step * steps / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "CG half sweep: every row-class launch of rank 0's least_squares calls",
                    "scope": "whole step on rank 0's shard (compute only)", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": ms / n, "algorithmic_bytes_per_step": per_step,
                    "timing_source": "HIP events around each least_squares call inside the timed region, rank 0"}

        result = sharded.bench(args, gpu, SHAPES, FACTORS, REG, CG_STEPS, rank0_roofline)
        if rank == 0:
            emit(saved_stdout, compact_sharded_line(result))
        return

    # ---- single GPU ---------------------------------------------------------------------------------
    t0 = time.time()
    Cui = synthetic_csr(users, items, nnz_target, gamma=gamma, seed=42)
    Ciu = Cui.T.tocsr()
    rng = np.random.default_rng(7)
    X0 = rng.random((users, FACTORS), dtype=np.float32) * 0.01
    Y0 = rng.random((items, FACTORS), dtype=np.float32) * 0.01
    t_gen = time.time() - t0

    t0 = time.time()
    Cui_d, Ciu_d = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    X, Y = gpu.Matrix(X0), gpu.Matrix(Y0)
    gram = gpu.Matrix.zeros(FACTORS, FACTORS)
    solver = gpu.LeastSquaresSolver()
    gpu.synchronize()
    t_upload = time.time() - t0

    def step():
        if args.solver == "cg":
            solver.calculate_yty(Y, gram, REG)
            solver.least_squares(Cui_d, X, gram, Y, CG_STEPS)
            solver.calculate_yty(X, gram, REG)
            solver.least_squares(Ciu_d, Y, gram, X, CG_STEPS)
        else:
            solver.calculate_yty(Y, gram, 0.0)
            solver.least_squares_cholesky(Cui_d, X, gram, Y, REG)
            solver.calculate_yty(X, gram, 0.0)
            solver.least_squares_cholesky(Ciu_d, Y, gram, X, REG)

    # The timed loop is fit()'s loop (implicit_amd/gpu/als.py): the four solver calls of an iteration are queued back to back
    # (deferred mode) and the host waits ONCE per iteration -- what the product does; a host round trip after each of the
    # four calls leaves the device idle for ~25 us each (2-3 % of a 4 ms step).  IMP_BENCH_SYNC_CALLS=1: synchronous calls.
    deferred = os.environ.get("IMP_BENCH_SYNC_CALLS") is None
    raw_step = step

    def step():  # noqa: F811
        raw_step()
        if deferred:
            gpu.synchronize()

    clock_idle = round(gpu.core_clock_mhz(50), 1)
    gpu.set_deferred_sync(deferred)
    for _ in range(args.warmup):
        step()
    gpu.synchronize()
    # HIP-event pairs cost stream time (~0.2 ms per iteration for all ~30 launches), so the timed region carries them
    # only for the dominant kernel family -- the mid-row team kernels of the CG sweep -- which is what `roofline`
    # reports; the per-kernel breakdown comes from a separate, untimed pass below.
    timed_filter = "als_cg_half_sweep" if args.solver == "cg" else None
    gpu.Profiler.reset()
    gpu.Profiler.enable(os.environ.get("IMP_BENCH_NO_PROF") is None, only=timed_filter)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gpu.synchronize()
    elapsed = time.perf_counter() - t0
    gpu.Profiler.enable(False)
    timed = {name: gpu.Profiler.get(name) for name in gpu.Profiler.names()}

    ms_per_step = 1e3 * elapsed / args.steps
    value = (users + items) / (elapsed / args.steps)

    detail_steps = min(args.steps, 3)
    gpu.Profiler.reset()
    gpu.Profiler.enable(True)
    for _ in range(detail_steps):
        step()
    gpu.synchronize()
    gpu.Profiler.enable(False)
    # the shader clock the steps run at: a one-wavefront probe queued right behind a step (the pool's boxes differ by a few per
    # cent in step time; this is the first thing to look at)
    clock_mhz = []
    for _ in range(3):
        raw_step()
        clock_mhz.append(round(gpu.core_clock_mhz(50), 1))
    # ... and the clock DURING the steps: the probe on a side stream beside six queued steps (still deferred mode)
    clock_beside = []
    for _ in range(3):
        for _ in range(6):
            raw_step()
        time.sleep(0.004)
        clock_beside.append(round(gpu.core_clock_mhz(-1500), 1))
        gpu.synchronize()
    gpu.set_deferred_sync(False)

    # ---- roofline: the WHOLE step (every row class of both half sweeps), per-class table as an extra ---------------
    cbytes = class_bytes_per_iteration(Cui, Ciu, FACTORS)
    kernels = {}
    for name in gpu.Profiler.names():
        ms, n = gpu.Profiler.get(name)
        kernels[name] = {"total_ms": ms, "launches": n}
    traffic, traffic_src = (None, None)
    if (args.shape, args.scale, args.solver, FACTORS) == ("lastfm360k", 1.0, "cg", 128):
        traffic, traffic_src = pmc_traffic_per_half_sweep(CG_STEPS)
    classes = {}
    for cname, knames in CLASS_KERNELS.items():
        ms = sum(kernels.get(k, {"total_ms": 0})["total_ms"] for k in knames)
        launches = sum(kernels.get(k, {"launches": 0})["launches"] for k in knames)
        if ms > 0:
            classes[cname] = {"ms_per_step": ms / detail_steps, "launches_per_step": launches / detail_steps,
                              "algorithmic_GB_per_step": cbytes[cname] / 1e9,
                              "achieved_GBps": cbytes[cname] * detail_steps / (ms * 1e-3) / 1e9,
                              "frac": cbytes[cname] * detail_steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "pmc_traffic_GB_per_step": (2 * traffic[cname] / 1e9) if traffic and cname in traffic else None}
    total_bytes = sum(cbytes.values())
    roofline = None
    sweep_ms, sweep_n = timed.get("als_cg_half_sweep", (0.0, 0))
    if args.solver == "cg":
        # `achieved` is quoted on the same clock as `value`: algorithmic bytes of one step / ms_per_step (gramians, launch
        # gaps and the host loop included); the HIP events around the two half sweeps of a step are the cross-check
        achieved = total_bytes / (elapsed / args.steps) / 1e9
        roofline = {"bound": "hbm", "kernel": "CG half sweep: every row-class launch of one least_squares call ("
                    + ", ".join(k for ks in CLASS_KERNELS.values() for k in ks if k in kernels) + ")",
                    "scope": "whole step = user half sweep + item half sweep, all row classes",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": (2 * sum(traffic.values())) if traffic else None, "traffic_source": traffic_src,
                    "traffic_note": "HBM bytes per STEP (both half sweeps) from the committed rocprofv3 PMC summary",
                    "algorithmic_bytes_per_step": total_bytes,
                    "avg_launch_ms": (sweep_ms / sweep_n) if sweep_n else None,
                    "half_sweeps_timed": sweep_n,
                    "frac_half_sweep_events": (total_bytes * args.steps / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sweep_ms else None,
                    "timing_source": "ms_per_step of the timed region (wall clock, synchronised both sides; inside it the "
                                     + ("four solver calls of a step are queued and waited for once, as fit() does" if deferred
                                        else "host waits after every solver call (IMP_BENCH_SYNC_CALLS)") +
                                     "); avg_launch_ms / frac_half_sweep_events: one HIP-event pair around each half sweep "
                                     "inside the same timed region",
                    "note": "algorithmic bytes = nnz(4f+8) + rows(8f+8) + 4f^2 per half sweep (SURVEY 8d); per-class "
                            "figures in `row_classes` (separate pass after the timed region, one event pair per launch)"}
    iteration_roofline = {"algorithmic_GB_per_step": total_bytes / 1e9,
                          "achieved_GBps": total_bytes / (elapsed / args.steps) / 1e9,
                          "frac_of_8TBps": total_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS}

    out = {
        "metric": "ALS user+item updates/sec per iteration (factors=128); top-k recs/sec",
        "value": value,
        "unit": "updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: last.fm-360K-shaped synthetic CSR, ALS CG cg_steps={CG_STEPS}"
                        if (args.shape, args.scale, args.solver, FACTORS) == ("lastfm360k", 1.0, "cg", 128)
                        else f"{args.shape} x{args.scale} f={FACTORS} {args.solver} (not the headline configuration)",
            "users": users, "items": items, "nnz": int(Cui.nnz), "factors": FACTORS, "regularization": REG,
            "solver": args.solver, "cg_steps": CG_STEPS if args.solver == "cg" else None, "parallelism": "1 GPU",
        },
        "nnz_visits_per_s": 2 * int(Cui.nnz) / (elapsed / args.steps),
        "roofline": roofline,
        "iteration_roofline": iteration_roofline,
        "row_classes": classes,
        "kernels_ms_per_step": {k: v["total_ms"] / detail_steps for k, v in kernels.items()},
        "kernels_note": f"per-kernel HIP-event times from {detail_steps} extra iterations after the timed region",
        "setup_s": {"generate": t_gen, "upload": t_upload},
        "core_clock_mhz": {"behind_a_step": clock_mhz, "beside_queued_steps": clock_beside, "before_warmup": clock_idle,
                           "note": "imp_debug_core_clock: core cycles over 50 us of the constant-rate wall clock, one wavefront "
                                   "queued behind the work named"},
    }

    if not args.no_topk:
        out["topk"] = bench_topk(gpu, Cui, X, Y, k=10)
        if not args.no_cpu_baseline:
            out["topk"]["cpu_baseline"] = cpu_topk_baseline(Cui, X, Y, k=10)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(Cui, Ciu, X0, Y0, args.cpu_seconds)
    headline = (args.shape, args.scale, args.solver, FACTORS) == ("lastfm360k", 1.0, "cg", 128)
    if headline and not args.no_extras:
        # secondary objects: the other BASELINE configurations, measured the same way (none of them is `value`)
        del Cui_d, Ciu_d, X, Y
        out["extras_note"] = ("secondary measurements on the other BASELINE configs (synthetic, inputs resident, HIP-event / "
                              "wall times of 3 iterations after 1 warm-up); `value` above is configs[2] only")
        for name, fn in (("fit_c3", lambda: extra_fit(gpu, Cui)), ("fp16_c3", lambda: extra_fp16(gpu, Cui, Ciu, X0, Y0)),
                         ("factor_grid", lambda: extra_factor_grid(gpu, Cui, Ciu)),
                         ("cholesky_f128", lambda: extra_cholesky_f128(gpu, Cui, Ciu)),
                         ("c1", lambda: extra_c1(gpu)),
                         ("c2", lambda: extra_c2(gpu, SHAPES)),
                         ("c5", lambda: extra_c5(gpu, SHAPES)), ("c4", lambda: extra_c4(gpu, SHAPES))):
            t0 = time.time()
            try:
                out.update(fn())
            except Exception as e:  # an extra must never cost the headline line
                out[name + "_error"] = f"{type(e).__name__}: {e}"
            out.setdefault("extras_s", {})[name] = round(time.time() - t0, 1)
    emit(saved_stdout, compact_line(out, args))


def compact_line(full, args):
    """The ONE stdout line.  The driver's record keeps the last 2000 characters of stdout and, of the parsed line, the contract
    keys + `roofline` + `cpu_baseline` + `config` (strings cut at 120 characters): so the line carries both halves of
    BASELINE.json's metric as short top-level keys, short strings only, one [ms, roofline fraction] pair per secondary
    configuration -- and every table (per-kernel times, row classes, PMC traffic per class, notes) goes to a side file:
    gpurun_out/bench_detail.json (override: IMP_BENCH_DETAIL=<path>; profiles/<round>_bench_detail.json is a committed copy)."""
    detail_path = os.environ.get("IMP_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {detail_path}: {e}\n")
        detail_path = None
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    topk = full.get("topk")
    if topk:
        # recs/s is quoted through model.recommend(userids, user_items[userids], N=k) with the liked-items filter
        line["topk_recs_per_s"] = _r(topk.get("value"))
    c = full["config"]
    line["config"] = {"workload": c["workload"], "users": c["users"], "items": c["items"], "nnz": c["nnz"], "factors": c["factors"],
                      "solver": c["solver"], "cg_steps": c["cg_steps"], "topk": "recommend() k=10, 20000 users, batches of 1000, liked-items filter" if topk else None}
    r = full.get("roofline")
    if r:
        line["roofline"] = {"bound": r["bound"], "achieved": _r(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _r(r["frac"], 4),
                            "traffic": _r(r["traffic"]), "kernel": "CG half sweep (all row-class launches)",
                            "avg_launch_ms": _r(r["avg_launch_ms"], 4), "frac_half_sweep_events": _r(r["frac_half_sweep_events"], 4),
                            "bytes_per_step": r["algorithmic_bytes_per_step"],
                            "classes_frac": {k: _r(v["frac"], 3) for k, v in full.get("row_classes", {}).items()},
                            "split_precision": "rows>512 nnz: fp16x2 MFMA normal matrix + fp32 fix-up; <=32: bf16x3 gramian; else fp32"}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "cores_total": os.cpu_count(),
                                "kind": cb["kind"], "sample": cb["sample_short"]}
        tcb = (topk or {}).get("cpu_baseline")
        if tcb:
            line["cpu_baseline"]["topk_recs_per_s"] = _r(tcb["value"])
            line["cpu_baseline"]["topk_cores"] = tcb["cores"]
    if topk:
        tr = topk.get("roofline") or {}
        line["topk"] = {"k": topk.get("k"), "via": "model.recommend()", "knn_topk_recs_per_s": _r(topk.get("knn_topk_recs_per_s")),
                        "presliced_recs_per_s": _r(topk.get("model_recommend_presliced_recs_per_s")),
                        "gemm_ms": _r(tr.get("avg_launch_ms"), 4), "gemm_frac_of_mfma_peak": _r(tr.get("frac"), 3),
                        "gemm_peak_TFLOPs": _r(tr.get("peak")),  # 2500 / partial products: 1 (screening pass), 3 or 6
                        "gemm_traffic": _r(tr.get("traffic"))}
    extras = {}
    for key, v in full.items():
        if key in ("cg_c3_f32", "cg_c3_f192", "cholesky_c3_f100"):  # side file only: the line must stay under 2000 characters
            continue
        if isinstance(v, dict) and "roofline" in v and key not in ("topk",) and isinstance(v.get("roofline"), dict):
            ms = v.get("ms_per_iter", v.get("compute_ms_per_iter", v.get("ms_per_batch")))
            extras[key] = [_r(ms, 3), _r(v["roofline"].get("frac"), 3)]
        elif key.startswith("c1_") and isinstance(v, dict):  # configs[0]: [GPU ms, reference 1-thread ms, worst rel. Frobenius]
            extras[key] = [_r(v["ms_per_iter"], 3), _r(v["cpu_1thread_ms_per_iter"], 3),
                           float(f"{max(v['rel_frobenius_gpu_vs_cpu'].values()):.2g}")]
    if extras:
        line["extras_ms_frac"] = extras
    errs = [k for k in full if k.endswith("_error")]
    if errs:
        line["extras_errors"] = errs
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    line["value"], line["ms_per_step"] = _r(line["value"]), _r(line["ms_per_step"], 4)
    while len(json.dumps(line)) > 1950 and line.get("extras_ms_frac"):  # never longer than the driver's 2000-character tail
        line["extras_ms_frac"].popitem()
    return line


def compact_sharded_line(full):
    """The N > 1 line (and `--gpus 1 --shape c4`): contract keys, the roofline of rank 0's compute, how the step divides on
    rank 0 (own kernels against exposed exchange), the ranks RCCL really connected; tables go to the side file."""
    detail_path = os.environ.get("IMP_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", f"bench_detail_n{full['n_gpus']}.json"))
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError:
        detail_path = None
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _r(line["value"])
    c = full["config"]
    line["config"] = {"workload": c["workload"][:118], "users": c["users"], "items": c["items"], "nnz": c["nnz"],
                      "factors": c["factors"], "cg_steps": c["cg_steps"], "parallelism": c["parallelism"][:118]}
    r = full.get("roofline")
    if r:
        line["roofline"] = {"bound": r["bound"], "achieved": _r(r["achieved"]), "peak": r["peak"], "unit": r["unit"],
                            "frac": _r(r["frac"], 4), "traffic": r.get("traffic"), "scope": "rank 0's shard, compute only",
                            "avg_launch_ms": _r(r.get("avg_launch_ms"), 4)}
    line["rccl_ranks_seen"] = full.get("rccl_ranks_seen")
    line["rank0_compute_ms"] = _r(full.get("rank0_compute_ms_per_step"), 3)
    line["rank0_exposed_exchange_ms"] = _r(full.get("rank0_exposed_exchange_ms_per_step"), 3)
    line["exchange_GB_per_rank"] = _r(full.get("exchange_GB_received_per_rank_per_step"), 4)
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    return line


def _r(x, digits=None):
    """Rounded for the compact line (None and strings pass through)."""
    if not isinstance(x, (int, float)) or isinstance(x, bool):
        return x
    if digits is None:
        return float(f"{x:.5g}")
    return round(float(x), digits)


FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def _time_iterations(gpu, step, iters=3, warmup=1):
    for _ in range(warmup):
        step()
    gpu.synchronize()
    gpu.Profiler.reset()
    gpu.Profiler.enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    gpu.synchronize()
    wall = (time.perf_counter() - t0) / iters
    gpu.Profiler.enable(False)
    kernels = {n: gpu.Profiler.get(n)[0] / iters for n in gpu.Profiler.names()}
    return wall, kernels


def _iteration_bytes(Cui, Ciu, f):
    return sum(cg_algorithmic_bytes(np.diff(M.indptr)[np.diff(M.indptr) > 0], f) for M in (Cui, Ciu))


def extra_fit(gpu, Cui):
    """model.fit() end to end on configs[2] (SURVEY 8d's method: the fit() path itself -- host transpose, CSR upload with
    the schedule build, iterations through the callback), 2 iterations."""
    from implicit_amd.als import AlternatingLeastSquares

    times = []
    model = AlternatingLeastSquares(factors=FACTORS, regularization=REG, iterations=2, random_state=1, use_gpu=True)
    t0 = time.time()
    model.fit(Cui, show_progress=False, callback=lambda it, dt, loss: times.append(dt))
    total = time.time() - t0
    return {"fit_c3": {"iterations": 2, "fit_s": total, "iteration_ms": [1e3 * t for t in times],
                       "setup_s": total - sum(times),
                       "note": "AlternatingLeastSquares.fit() on the configs[2] matrix: setup = float32/CSR checks, host "
                               "transpose, two CSRMatrix uploads with their row schedules, factor init and upload"}}


def extra_fp16(gpu, Cui, Ciu, X0, Y0):
    """configs[2] with fp16 factor STORAGE (the reference's dtype=np.float16 mode, tests/als_test.py:30-34): the f=128 kernels
    load / store half precision directly, arithmetic and CG state stay fp32 -- the gathered bytes per nonzero halve."""
    f = X0.shape[1]
    X, Y = gpu.Matrix(X0.astype(np.float16)), gpu.Matrix(Y0.astype(np.float16))
    Cd, Ctd = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()

    def step():
        solver.calculate_yty(Y, gram, REG)
        solver.least_squares(Cd, X, gram, Y, CG_STEPS)
        solver.calculate_yty(X, gram, REG)
        solver.least_squares(Ctd, Y, gram, X, CG_STEPS)

    t, kernels = _time_iterations(gpu, step)
    rows, nnz = Cui.shape[0] + Cui.shape[1], int(Cui.nnz)
    gb = (2 * nnz * (2 * f + 8) + rows * (4 * f + 8) + 2 * 4 * f * f) / 1e9  # SURVEY 8d's formula with 2-byte factors
    model_rows = 1000
    fold = _fold_in_latency(gpu, Cui, X0, Y0, model_rows)
    return {"fp16_c3": {"workload": "configs[2] matrix, factors stored as float16 (fp32 arithmetic), CG cg_steps=%d" % CG_STEPS,
                        "ms_per_iter": 1e3 * t, "updates_per_s": rows / t,
                        "roofline": {"bound": "hbm", "achieved": gb / t, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": gb / t / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                        "kernels_ms_per_iter": kernels},
            "fold_in_c3": fold}


def _fold_in_latency(gpu, Cui, X0, Y0, batch):
    """recalculate_user / partial_fit_users (implicit/gpu/als.py:184-278) on a trained-shape model: the second caller of the
    solver, small batches where launch latency and the CSR upload dominate."""
    from implicit_amd.als import AlternatingLeastSquares

    model = AlternatingLeastSquares(factors=X0.shape[1], regularization=REG, use_gpu=True)
    model.user_factors, model.item_factors = gpu.Matrix(X0), gpu.Matrix(Y0)
    out = {}
    for n in (1, 100, batch):
        ids = np.arange(n)
        rows = Cui[:n]
        model.recalculate_user(ids, rows)  # warm-up (builds the cached gramian)
        gpu.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            model.recalculate_user(ids, rows)
        gpu.synchronize()
        out[f"recalculate_user_{n}_rows_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    out["note"] = "host CSR slice upload + one solver call per batch; Cholesky fold-in for factors <= 160, else CG run to f steps"
    return out


def extra_factor_grid(gpu, Cui, Ciu):
    """The reference's published factor grid (benchmarks/README.md:29-32,51-58: factors 32 / 64 / 128 / 192 / 256 on last.fm-360K)
    on the configs[2] matrix, CG cg_steps = 3, fp32: one object per factor count, each with its roofline."""
    out = {}
    rows = Cui.shape[0] + Cui.shape[1]
    Cd, Ctd = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    solver = gpu.LeastSquaresSolver()
    rng = np.random.default_rng(7)
    for f in (32, 64, 192, 256):
        X = gpu.Matrix(rng.random((Cui.shape[0], f), dtype=np.float32) * 0.01)
        Y = gpu.Matrix(rng.random((Cui.shape[1], f), dtype=np.float32) * 0.01)
        gram = gpu.Matrix.zeros(f, f)

        def cg():
            solver.calculate_yty(Y, gram, REG)
            solver.least_squares(Cd, X, gram, Y, CG_STEPS)
            solver.calculate_yty(X, gram, REG)
            solver.least_squares(Ctd, Y, gram, X, CG_STEPS)

        t, kernels = _time_iterations(gpu, cg)
        gb = _iteration_bytes(Cui, Ciu, f) / 1e9
        out[f"cg_c3_f{f}"] = {"workload": f"configs[2] matrix, factors={f}, CG cg_steps={CG_STEPS}", "ms_per_iter": 1e3 * t,
                              "updates_per_s": rows / t,
                              "roofline": {"bound": "hbm", "achieved": gb / t, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": gb / t / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                              "kernels_ms_per_iter": kernels}
        del X, Y, gram
    return out


def extra_cholesky_f128(gpu, Cui, Ciu):
    """The Cholesky solver at the factor count the metric is quoted on (`use_cg=False`, implicit/cpu/als.py:418-423) on the
    configs[2] matrix: since round 5 the rows' normal matrices are built on the matrix cores and factorised on their LDS images
    (als_cg_nm.hip nm_chol; IMP_CHOL_NM=0: the workgroup-per-row LDS kernel of round 4) -- and at f = 100, the reference's CPU
    default, which rides the same path zero-padded (IMP_CHOL_PAD=0: the workgroup kernel)."""
    out = {}
    Cd, Ctd = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    solver = gpu.LeastSquaresSolver()
    rows = Cui.shape[0] + Cui.shape[1]
    for f, key, what in ((FACTORS, "cholesky_c3_f128", "normal matrices on the matrix cores, LDS-image factorisation"),
                         (100, "cholesky_c3_f100", "zero-padded onto the f=128 path")):
        rng = np.random.default_rng(7)
        X = gpu.Matrix(rng.random((Cui.shape[0], f), dtype=np.float32) * 0.01)
        Y = gpu.Matrix(rng.random((Cui.shape[1], f), dtype=np.float32) * 0.01)
        gram = gpu.Matrix.zeros(f, f)

        def chol():
            solver.calculate_yty(Y, gram, 0.0)
            solver.least_squares_cholesky(Cd, X, gram, Y, REG)
            solver.calculate_yty(X, gram, 0.0)
            solver.least_squares_cholesky(Ctd, Y, gram, X, REG)

        t, kernels = _time_iterations(gpu, chol, iters=2)
        flops = 2.0 * Cui.nnz * 2 * f * f + rows * (f ** 3 / 3.0 + 2.0 * f * f)   # of the f the caller asked for, not the padded one
        out[key] = {"workload": f"configs[2] matrix, f={f}, Cholesky ({what})", "ms_per_iter": 1e3 * t,
                    "updates_per_s": rows / t, "tflops": flops / t / 1e12,
                    "roofline": {"bound": "fp32", "achieved": flops / t / 1e12, "peak": FP32_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": flops / t / 1e12 / FP32_PEAK_TFLOPS},
                    "kernels_ms_per_iter": kernels}
        del X, Y, gram
    return out


def extra_c1(gpu):
    """BASELINE configs[0]: MovieLens-100K shape (943 x 1682, ~100 K nnz), f = 16 -- the reference's own CPU-runnable case
    (1 CPU thread, plumbing): the compiled reference (oracle/_ref, else the plain-C port) timed on ONE thread beside the GPU on
    the same inputs, CG cg_steps=3 and Cholesky, with the relative Frobenius distance of one iteration from identical factors."""
    from implicit_amd.synthetic import named
    from oracle import oracle as port
    from oracle import ref

    f = 16
    C = named("ml100k")
    Ct = C.T.tocsr()
    rng = np.random.default_rng(7)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.1 - 0.05
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.1 - 0.05
    als_ref, _ = ref.load()
    if als_ref is None:
        port.build()
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    solver = gpu.LeastSquaresSolver()
    gram = gpu.Matrix.zeros(f, f)
    rows = C.shape[0] + C.shape[1]
    out = {}
    for solver_name in ("cg", "cholesky"):
        X, Y = gpu.Matrix(X0), gpu.Matrix(Y0)

        def gpu_iter():
            if solver_name == "cg":
                solver.calculate_yty(Y, gram, REG)
                solver.least_squares(Cd, X, gram, Y, CG_STEPS)
                solver.calculate_yty(X, gram, REG)
                solver.least_squares(Ctd, Y, gram, X, CG_STEPS)
            else:
                solver.calculate_yty(Y, gram, 0.0)
                solver.least_squares_cholesky(Cd, X, gram, Y, REG)
                solver.calculate_yty(X, gram, 0.0)
                solver.least_squares_cholesky(Ctd, Y, gram, X, REG)

        def cpu_iter(Xh, Yh):
            if als_ref is not None:
                if solver_name == "cg":
                    als_ref.least_squares_cg(C, Xh, Yh, REG, num_threads=1, cg_steps=CG_STEPS)
                    als_ref.least_squares_cg(Ct, Yh, Xh, REG, num_threads=1, cg_steps=CG_STEPS)
                else:
                    als_ref.least_squares(C, Xh, Yh, REG, num_threads=1)
                    als_ref.least_squares(Ct, Yh, Xh, REG, num_threads=1)
            elif solver_name == "cg":
                port.least_squares_cg(C, Xh, Yh, REG, num_threads=1, cg_steps=CG_STEPS)
                port.least_squares_cg(Ct, Yh, Xh, REG, num_threads=1, cg_steps=CG_STEPS)
            else:
                port.least_squares(C, Xh, Yh, REG, num_threads=1)
                port.least_squares(Ct, Yh, Xh, REG, num_threads=1)

        gpu_iter()  # one iteration from (X0, Y0) on both sides: parity
        Xh, Yh = X0.copy(), Y0.copy()
        cpu_iter(Xh, Yh)
        ex = float(np.linalg.norm(X.to_numpy() - Xh) / np.linalg.norm(Xh))
        ey = float(np.linalg.norm(Y.to_numpy() - Yh) / np.linalg.norm(Yh))
        t_gpu, _ = _time_iterations(gpu, gpu_iter, iters=20, warmup=2)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or time.perf_counter() - t0 < 0.5:
            cpu_iter(Xh, Yh)
            reps += 1
        t_cpu = (time.perf_counter() - t0) / reps
        out[f"c1_{solver_name}"] = {"workload": "BASELINE configs[0]: MovieLens-100K shape %d x %d, %d nnz, f=16, %s" %
                                                (C.shape[0], C.shape[1], C.nnz, solver_name),
                                    "ms_per_iter": 1e3 * t_gpu, "updates_per_s": rows / t_gpu,
                                    "cpu_1thread_ms_per_iter": 1e3 * t_cpu, "cpu_1thread_updates_per_s": rows / t_cpu,
                                    "cpu_kind": "reference" if als_ref is not None else "port",
                                    "rel_frobenius_gpu_vs_cpu": {"users": ex, "items": ey},
                                    "note": "launch-latency bound on the GPU (a 1 ms problem); listed because it is BASELINE's configs[0]"}
    return out


def extra_c2(gpu, SHAPES):
    """BASELINE configs[1]: 1M x 100K x 50M nnz, f = 64 -- Cholesky (the configuration's solver) and CG 3."""
    from implicit_amd.synthetic import named

    f = 64
    C = named("c2")
    Ct = C.T.tocsr()
    rng = np.random.default_rng(7)
    X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
    Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()

    def chol():
        solver.calculate_yty(Y, gram, 0.0)
        solver.least_squares_cholesky(Cd, X, gram, Y, REG)
        solver.calculate_yty(X, gram, 0.0)
        solver.least_squares_cholesky(Ctd, Y, gram, X, REG)

    def cg():
        solver.calculate_yty(Y, gram, REG)
        solver.least_squares(Cd, X, gram, Y, CG_STEPS)
        solver.calculate_yty(X, gram, REG)
        solver.least_squares(Ctd, Y, gram, X, CG_STEPS)

    rows = C.shape[0] + C.shape[1]
    t_chol, k_chol = _time_iterations(gpu, chol)
    flops = 2.0 * C.nnz * 2 * f * f + rows * (f ** 3 / 3.0 + 2.0 * f * f)  # DESIGN 4 (kernel table): nnz 2f^2 (SYRK) + R (f^3/3 + 2f^2), both sides
    X.copy_from_numpy(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
    Y.copy_from_numpy(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
    t_cg, k_cg = _time_iterations(gpu, cg)
    gb = _iteration_bytes(C, Ct, f) / 1e9
    return {"cholesky_c2": {"workload": "BASELINE configs[1]: 1M x 100K, %d nnz, f=64, Cholesky" % C.nnz, "ms_per_iter": 1e3 * t_chol,
                            "updates_per_s": rows / t_chol, "tflops": flops / t_chol / 1e12,
                            "roofline": {"bound": "fp32", "achieved": flops / t_chol / 1e12, "peak": FP32_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": flops / t_chol / 1e12 / FP32_PEAK_TFLOPS},
                            "kernels_ms_per_iter": k_chol},
            "cg_c2": {"workload": "same matrix, CG cg_steps=%d" % CG_STEPS, "ms_per_iter": 1e3 * t_cg, "updates_per_s": rows / t_cg,
                      "roofline": {"bound": "hbm", "achieved": gb / t_cg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gb / t_cg / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                      "kernels_ms_per_iter": k_cg}}


def extra_c5(gpu, SHAPES):
    """BASELINE configs[4]: MovieLens-20M shape, f = 256 fp32 CG + KnnQuery similar_items k = 100."""
    from implicit_amd.synthetic import named

    f = 256
    C = named("ml20m")
    Ct = C.T.tocsr()
    rng = np.random.default_rng(7)
    X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
    Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()

    def cg():
        solver.calculate_yty(Y, gram, REG)
        solver.least_squares(Cd, X, gram, Y, CG_STEPS)
        solver.calculate_yty(X, gram, REG)
        solver.least_squares(Ctd, Y, gram, X, CG_STEPS)

    t_cg, k_cg = _time_iterations(gpu, cg)
    rows = C.shape[0] + C.shape[1]
    gb = _iteration_bytes(C, Ct, f) / 1e9
    # similar_items(k=100): cosine top-k of item batches against all items (gpu/matrix_factorization_base.py:162-200)
    norms = gpu.calculate_norms(Y)
    knn = gpu.KnnQuery()
    batch, n_batches = 1000, 10
    views = [Y[b * batch:(b + 1) * batch] for b in range(n_batches)]
    knn.topk(Y, views[0], 100, item_norms=norms)
    gpu.synchronize()
    t0 = time.perf_counter()
    for v in views:
        knn.topk(Y, v, 100, item_norms=norms)
    gpu.synchronize()
    t = time.perf_counter() - t0
    gpu.Profiler.reset()
    gpu.Profiler.enable(True)
    for v in views:
        knn.topk(Y, v, 100, item_norms=norms)
    gpu.synchronize()
    gpu.Profiler.enable(False)
    sim_kernels = {n: gpu.Profiler.get(n)[0] / n_batches for n in gpu.Profiler.names()}
    gemm_ms = sim_kernels.get("score_gemm", 0.0)
    sim_flops = 2.0 * batch * Y.shape[0] * f
    sim_roofline = None
    if gemm_ms > 0:
        exact = os.environ.get("IMP_TOPK_FP32_MFMA") is not None
        resident = os.environ.get("IMP_TOPK_RESIDENT", "1") != "0" and not exact
        sim_peak = FP32_PEAK_TFLOPS if exact else BF16_PEAK_TFLOPS / (3.0 if resident else 6.0)
        sim_roofline = {"bound": "mfma", "kernel": "score_resident_kernel<16, 1, 0>" if resident else "score_gemm_direct_kernel<0",
                        "achieved": sim_flops / (gemm_ms * 1e-3) / 1e12,
                        "peak": sim_peak, "unit": "TFLOP/s", "frac": sim_flops / (gemm_ms * 1e-3) / 1e12 / sim_peak,
                        "avg_launch_ms": gemm_ms, "flops_per_launch": sim_flops, "traffic": None,
                        "note": "fp32-equivalent flops; peak = dense fp16 MFMA / 3 partial products of the two-term form (a 1000 x 26 744 x 256 "
                                "product is 16 us of matrix time: the launch is latency, the two select kernels beside it weigh more)"
                                if resident else "fp32-equivalent flops; peak = dense bf16 MFMA / 6 partial products"}
    return {"cg_c5": {"workload": "BASELINE configs[4]: 138,493 x 26,744, %d nnz, f=256 fp32, CG cg_steps=%d" % (C.nnz, CG_STEPS),
                      "ms_per_iter": 1e3 * t_cg, "updates_per_s": rows / t_cg,
                      "roofline": {"bound": "hbm", "achieved": gb / t_cg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gb / t_cg / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                      "kernels_ms_per_iter": k_cg},
            "similar_items_c5": {"workload": "KnnQuery similar_items k=100 over all 26,744 items with norms, batches of 1000 items, f=256",
                                 "items_per_s": batch * n_batches / t, "ms_per_batch": 1e3 * t / n_batches,
                                 "scoring_TFLOPs": 2.0 * batch * n_batches * Y.shape[0] * f / t / 1e12,
                                 "roofline": sim_roofline, "kernels_ms_per_batch": sim_kernels,
                                 "note": "ids/scores returned to host memory per batch"}}


def extra_c4(gpu, SHAPES):
    """BASELINE configs[3] (10M x 1M x 500M nnz, f = 128) on this one GPU, two ways:
      c4_full_1gpu  the WHOLE matrix (14 GB of the 288): the one-GPU point of the strong-scaling curve `bench.py --gpus N`
                    measures for N > 1 (same matrix: the 8 x 8 block grid does not depend on N);
      c4_shard      what rank 0 of the 8-GPU run computes per iteration -- its 1.25M user rows against the item replica
                    and its 125K item rows against the 10M-row user replica (no exchange: that needs the other seven)."""
    from implicit_amd.synthetic import grid_shards

    users, items, nnz, gamma = SHAPES["c4"]
    t0 = time.time()
    Cui, Ciu, _, _ = grid_shards(0, 1, users, items, nnz, 8, gamma=gamma, seed=42)
    t_gen = time.time() - t0
    f = FACTORS
    X = gpu.RandomState(7).uniform(users, f, 0.0, 0.01)
    Y = gpu.RandomState(8).uniform(items, f, 0.0, 0.01)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()
    out = {}

    t0 = time.time()
    Cd, Ctd = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    t_upload = time.time() - t0

    def full_step():
        solver.calculate_yty(Y, gram, REG)
        solver.least_squares(Cd, X, gram, Y, CG_STEPS)
        solver.calculate_yty(X, gram, REG)
        solver.least_squares(Ctd, Y, gram, X, CG_STEPS)

    t, kernels = _time_iterations(gpu, full_step, iters=2)
    gb = _iteration_bytes(Cui, Ciu, f) / 1e9
    out["c4_full_1gpu"] = {"workload": "BASELINE configs[3] whole on ONE GPU: %d x %d, %d nnz, f=128, CG cg_steps=%d (strong-scaling "
                                       "base of `bench.py --gpus N`)" % (users, items, Cui.nnz, CG_STEPS),
                           "ms_per_iter": 1e3 * t, "updates_per_s": (users + items) / t,
                           "roofline": {"bound": "hbm", "achieved": gb / t, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": gb / t / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                           "kernels_ms_per_iter": kernels, "generate_s": t_gen, "upload_and_schedule_s": t_upload}
    del Cd, Ctd

    # rank 0's eighth: rows [0, users/8) of Cui, rows [0, items/8) of Ciu (the grid's first block row / block column)
    nu, ni = users // 8, items // 8
    Cui_s, Ciu_s = Cui[:nu], Ciu[:ni]
    del Cui, Ciu
    Cd, Ctd = gpu.CSRMatrix(Cui_s), gpu.CSRMatrix(Ciu_s)
    Xm, Ym = X[0:nu], Y[0:ni]

    def shard_step():
        solver.calculate_yty(Ym, gram, REG)      # the rank's partial gramian (all-reduced over xGMI in the real run)
        solver.least_squares(Cd, Xm, gram, Y, CG_STEPS)
        solver.calculate_yty(Xm, gram, REG)
        solver.least_squares(Ctd, Ym, gram, X, CG_STEPS)

    t, kernels = _time_iterations(gpu, shard_step, iters=2)
    gb = _iteration_bytes(Cui_s, Ciu_s, f) / 1e9
    xgmi_gb = 7.0 / 8.0 * (users + items) * f * 4 / 1e9
    # modelled 8-GPU iteration (implicit_amd.gpu.sharded.project_iteration_ms): K chunks of falling size per half sweep, the
    # last chunk's exchange exposed, +15 % compute while RCCL's kernels are resident, one xGMI link per peer
    from implicit_amd.gpu import sharded as _sh

    half_ms = [1e3 * t / 2.0] * 2   # split evenly: the two half sweeps of rank 0's shard cost about the same
    recv = [7.0 / 8.0 * users * f * 4, 7.0 / 8.0 * items * f * 4]
    model = {}
    for link in (50.0, 75.0):
        ms, hidden = _sh.project_iteration_ms(half_ms, recv, 8, link_GBps=link)
        model[f"link_{int(link)}GBps"] = {"ms_per_iter": ms, "updates_per_s": (users + items) / (ms * 1e-3),
                                          "speedup_over_1gpu": out["c4_full_1gpu"]["ms_per_iter"] / ms}
    out["c4_shard"] = {"workload": "rank 0 of BASELINE configs[3] on 8 GPUs: %d user rows (%d nnz) + %d item rows (%d nnz), f=128, "
                                   "CG cg_steps=%d" % (Cui_s.shape[0], Cui_s.nnz, Ciu_s.shape[0], Ciu_s.nnz, CG_STEPS),
                       "compute_ms_per_iter": 1e3 * t,
                       "roofline": {"bound": "hbm", "achieved": gb / t, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": gb / t / HBM_PEAK_GBS, "algorithmic_GB_per_iter": gb},
                       "projected_8gpu_updates_per_s_if_exchange_hidden": (users + items) / t,
                       "projected_8gpu_model": model,
                       "projected_8gpu_model_note": "timeline model: %d chunks of falling size per half sweep (ratio 0.75), exchange of "
                                                    "chunk k beside the solve of chunk k+1, last chunk exposed, +15 %% compute under "
                                                    "resident RCCL kernels, one xGMI link per peer at the stated rate" % _sh.default_chunks(8),
                       "exchange_GB_received_per_rank_per_iter": xgmi_gb,
                       "kernels_ms_per_iter": kernels}
    return out


def cpu_topk_baseline(Cui, X, Y, k=10, seconds=8.0):
    """The reference's CPU scorer (implicit/cpu/topk.pyx:15-67 via oracle/_ref, else the plain-C port) on a bounded query
    sample of the same recommend()-shaped workload, all host threads (num_threads=0)."""
    from oracle import oracle as port
    from oracle import ref

    _, topk_ref = ref.load()
    items = Y.to_numpy()
    n = 64
    queries = X[0:4096].to_numpy()

    def run(q):
        t = time.time()
        if topk_ref is not None:
            liked = Cui[:len(q)]
            topk_ref.topk(items, q, k, None, liked, None, 0)
        else:
            port.topk(items, q, k)
        return time.time() - t

    t = run(queries[:n])
    n = int(min(len(queries), max(n, n * seconds / max(t, 1e-3))))
    t = run(queries[:n])
    return {"value": n / t, "unit": "recs/s", "cores": os.cpu_count(), "kind": "reference" if topk_ref is not None else "port",
            "sample": f"{n} queries x {items.shape[0]} items, k={k}, liked-items filter on, {t:.1f}s, num_threads=0"}


def bench_topk(gpu, Cui, X, Y, k=10, queries=20_000, batch=1000):
    """recommend()-shaped scoring: top-k over all items for `queries` users in batches of 1000 with the
    liked-items filter active (examples/lastfm.py:150-156 of the reference)."""
    knn = gpu.KnnQuery()
    queries = min(queries, Cui.shape[0])
    filt = [gpu.COOMatrix(Cui[s:s + batch].tocoo()) for s in range(0, queries, batch)]
    views = [X[s:min(s + batch, queries)] for s in range(0, queries, batch)]
    knn.topk(Y, views[0], k, query_filter=filt[0])  # warm-up
    gpu.synchronize()
    gpu.Profiler.enable(False)  # the HIP-event pairs cost ~0.4 ms per call here: timed without them
    t0 = time.perf_counter()
    for v, fl in zip(views, filt):
        knn.topk(Y, v, k, query_filter=fl)
    gpu.synchronize()
    t = time.perf_counter() - t0
    gpu.Profiler.reset()  # per-kernel times from a second, untimed pass
    gpu.Profiler.enable(True)
    for v, fl in zip(views, filt):
        knn.topk(Y, v, k, query_filter=fl)
    gpu.synchronize()
    gpu.Profiler.enable(False)
    kernels = {name: gpu.Profiler.get(name)[0] / len(views) for name in gpu.Profiler.names()}
    # the shader clock behind a run of scoring launches (a one-wavefront probe queued right behind them): the matrix-core peak is
    # quoted at 2.4 GHz, a power-limited box runs an MFMA-heavy kernel below that
    topk_clock = []
    for _ in range(2):
        for v, fl in zip(views[:8], filt[:8]):
            knn.topk(Y, v, k, query_filter=fl)
        topk_clock.append(round(gpu.core_clock_mhz(50), 1))
    flops = 2.0 * queries * Y.shape[0] * Y.shape[1]
    # the dominant kernel: the full scoring GEMM (emit epilogue), HIP events of the profiled pass
    gemm_ms = kernels.get("score_gemm", 0.0)
    per_batch_flops = 2.0 * batch * Y.shape[0] * Y.shape[1]
    roofline = None
    if gemm_ms > 0:
        tf = per_batch_flops / (gemm_ms * 1e-3) / 1e12
        exact = os.environ.get("IMP_TOPK_FP32_MFMA") is not None
        resident = os.environ.get("IMP_TOPK_RESIDENT", "1") != "0" and not exact and Y.shape[1] <= 256
        screen = resident and os.environ.get("IMP_TOPK_SCREEN", "1") != "0"
        products = 1.0 if screen else (3.0 if resident else 6.0)
        peak = FP32_PEAK_TFLOPS if exact else BF16_PEAK_TFLOPS / products
        kernel = ("score_resident_kernel<8, 2, 3>" if screen else "score_resident_kernel<8, 2, 2>") if resident else "score_gemm_direct_kernel<2"
        roofline = {"bound": "mfma", "kernel": kernel + " (emit pass)", "achieved": tf, "peak": peak,
                    "unit": "TFLOP/s", "frac": tf / peak, "avg_launch_ms": gemm_ms,
                    "flops_per_launch": per_batch_flops, "traffic": _pmc_kernel_traffic(kernel),
                    "traffic_note": "HBM bytes per launch of the emit GEMM from the committed rocprofv3 PMC summary "
                                    "(FETCH_SIZE corrected for gfx950); algorithmic operand bytes = items x f x 4 once per launch",
                    "note": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); 2 x batch x items x f flops per launch" if exact else
                            ("fp32-equivalent flops (2 x batch x items x f per launch); the product runs as %d partial products of %s on "
                             "the fp16 / bf16 matrix cores with fp32 accumulation: peak = 2500 TFLOP/s dense / %d" %
                             (int(products), ("the high fp16 planes only: a SCREENING pass against tau - eps (rigorous error bound); the select kernel "
                                              "re-scores the few entries that can still be among the best k in fp32 from the stored factors") if screen
                              else ("two-term fp16 operands (l h, h l, h h; per-row query scales, exact item maximum)" if resident
                                    else "three-term bf16 operands"), int(products)))}
    # the model-level call a user makes (recommend(): host COO build of the liked items + upload + KnnQuery.topk per batch)
    rec, rec_presliced = None, None
    try:
        from implicit_amd.als import AlternatingLeastSquares

        model = AlternatingLeastSquares(factors=Y.shape[1], use_gpu=True)
        model.user_factors, model.item_factors = X, Y
        ids = np.arange(queries)
        model.recommend(ids[:batch], Cui[:batch], N=k)
        gpu.synchronize()
        t0 = time.perf_counter()
        for s0 in range(0, queries, batch):
            model.recommend(ids[s0:s0 + batch], Cui[s0:s0 + batch], N=k)
        gpu.synchronize()
        rec = queries / (time.perf_counter() - t0)
        # the same calls with the caller's row slices prepared beforehand: user_items[a:b] is scipy's row slice in the CALLER
        # (0.13 ms per 1000-row batch of this matrix: four fifths of everything recommend() costs on the host)
        slices = [Cui[s0:s0 + batch] for s0 in range(0, queries, batch)]
        gpu.synchronize()
        t0 = time.perf_counter()
        for j, s0 in enumerate(range(0, queries, batch)):
            model.recommend(ids[s0:s0 + batch], slices[j], N=k)
        gpu.synchronize()
        rec_presliced = queries / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        rec = f"{type(e).__name__}: {e}"
    # recs/s is quoted through the model-level call; the raw KnnQuery.topk rate (filters already resident) is kept beside it
    value = rec if isinstance(rec, float) else queries / t
    return {"metric": "top-k recs/sec", "value": value, "unit": "recs/s", "k": k, "queries": queries,
            "via": "AlternatingLeastSquares.recommend(userids, user_items[userids], N=10), filter_already_liked_items=True"
                   if isinstance(rec, float) else "KnnQuery.topk (model.recommend failed: %s)" % rec,
            "knn_topk_recs_per_s": queries / t, "model_recommend_recs_per_s": rec,
            "model_recommend_presliced_recs_per_s": rec_presliced, "core_clock_mhz_behind_scoring": topk_clock,
            "kernels_ms_per_batch": kernels, "scoring_TFLOPs": flops / t / 1e12, "roofline": roofline,
            "batch": batch, "items": Y.shape[0], "filter_already_liked_items": True,
            "ids": "identical to the compiled reference's topk outside fp32 near-ties (PARITY.md)",
            "note": "value: AlternatingLeastSquares.recommend() for 20 000 users in batches of 1000 -- the caller's scipy row slice "
                    "user_items[a:b], the host COO pattern + upload, KnnQuery.topk, ids/scores back in host memory; "
                    "knn_topk_recs_per_s: KnnQuery.topk alone with the liked-items COO filters already on the device; kernel times "
                    "from a separate profiled pass"}


if __name__ == "__main__":
    main()
