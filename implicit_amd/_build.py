"""Builds libimplicit_hip.so (hipcc, gfx950) in-tree next to this file.

`python -m implicit_amd._build [--force]`.  Each .hip translation unit is compiled to an object in
parallel and linked into implicit_amd/libimplicit_hip.so, which links librccl (multi-GPU exchange)
and the HIP runtime only -- no PyTorch, no BLAS/solver/rand vendor libraries.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libimplicit_hip.so")
SOURCES = ["containers.hip", "als_cg.hip", "als_cg_q.hip", "als_cg_qf.hip", "als_cg_qh.hip", "als_cg_fixup.hip", "als_cg_nm.hip", "als_cg_w256.hip", "als_cholesky.hip", "gramian.hip", "solver.hip", "topk.hip",
           "random.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-pass-failed", "-ffp-contract=off"]
# per-file additions.  als_cholesky.hip: the SLP vectoriser packs the independent accumulators of the unrolled factorisation into
# v_pk_fma_f32 pairs that have to be assembled with v_mov storms (1.7 K of them) and doubles the register count
EXTRA_FLAGS = {"als_cholesky.hip": ["-fno-slp-vectorize"]}


def _deps():
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "implicit_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_mtime):
        return obj, False
    subprocess.check_call([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", path, "-o", obj])
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_mtime = _deps()
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr_mtime), SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB,
                               "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
        if verbose:
            print(f"[implicit_amd] built {LIB}")
    return LIB


def build_variant(tag, defines, sources=("als_cg_qf.hip",)):
    """A/B aid: the library with `sources` recompiled under extra -D flags, written to build/variants/libimplicit_hip_<tag>.so
    (git-ignored; it travels to the GPU box) and selected at run time with IMP_LIB_PATH.  Every other object is shared with
    the regular build."""
    build(verbose=False)
    out_dir = os.path.join(HERE, "..", "build", "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if src in sources:
            obj = os.path.join(out_dir, f"{tag}_{src.replace('.hip', '.o')}")
            subprocess.check_call([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), *[f"-D{d}" for d in defines], "-c",
                                   os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    lib = os.path.abspath(os.path.join(out_dir, f"libimplicit_hip_{tag}.so"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib, "-L/opt/rocm/lib", "-lrccl",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m implicit_amd._build --variant TAG [--sources a.hip,b.hip] DEFINE [DEFINE ...]
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        sources = ("als_cg_qf.hip",)
        if rest and rest[0] == "--sources":
            sources, rest = tuple(rest[1].split(",")), rest[2:]
        print(build_variant(sys.argv[i + 1], rest, sources))
    else:
        build(force="--force" in sys.argv)
