#!/usr/bin/env python3
"""Acceptance mode (A) of SURVEY.md section 7 / 8c: the REFERENCE's own package and tests over this repository's shim.

TEST INFRASTRUCTURE ONLY (as everything under oracle/).  Assembles, in the git-ignored build/refsuite/ tree:

  implicit/            the reference's Python sources, copied from /root/reference/implicit (*.py only) -- its model
                       layer (implicit/gpu/als.py, matrix_factorization_base.py, implicit/als.py ...) is what runs
  implicit/cpu/*.so    the reference's CPU extensions compiled by oracle/build_ref.py (oracle/_ref)
  implicit/gpu/_cuda.py   ONE line: `from implicit_amd.gpu._cuda import *` -- the drop-in point: where the reference
                       loads its CUDA extension (implicit/gpu/__init__.py:15) it gets the ctypes shim over
                       libimplicit_hip.so instead
  rmm/                 empty stub (the reference imports rmm before its extension, implicit/gpu/__init__.py:11)
  tests/               the reference's tests/{als_test,gpu_test,recommender_base_test}.py, unmodified

Nothing of this enters the git history (build/ is ignored); the tree travels to the GPU box with the snapshot, where
tests/test_reference_suite.py runs the reference tests under `-m gpu`.  A no-op when /root/reference is absent.
"""
import glob
import os
import shutil
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "build", "refsuite")
TESTS = ["als_test.py", "gpu_test.py", "recommender_base_test.py"]
# extension module -> package directory inside implicit/
EXT_PLACES = {"_als": "cpu", "topk": "cpu", "bpr": "cpu", "lmf": "cpu", "evaluation": "", "_nearest_neighbours": ""}


def assemble(reference="/root/reference", verbose=True):
    if not os.path.isdir(reference):
        return os.path.isdir(OUT)
    from oracle import build_ref

    if not build_ref.build(reference, verbose=verbose, extra=True):
        return False
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    pkg = os.path.join(OUT, "implicit")
    for sub in ("", "cpu", "gpu", "ann", "datasets"):
        src = os.path.join(reference, "implicit", sub)
        dst = os.path.join(pkg, sub)
        os.makedirs(dst, exist_ok=True)
        for f in glob.glob(os.path.join(src, "*.py")):
            shutil.copy(f, dst)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for name, sub in EXT_PLACES.items():
        shutil.copy(os.path.join(build_ref.OUT, name + suffix), os.path.join(pkg, sub))
    with open(os.path.join(pkg, "gpu", "_cuda.py"), "w") as f:
        f.write('"""Drop-in point: the reference loads its CUDA extension here (implicit/gpu/__init__.py:15)."""\n'
                "from implicit_amd.gpu._cuda import *  # noqa: F401,F403\n"
                "from implicit_amd.gpu._cuda import Matrix, KnnQuery  # noqa: F401  (named by tests/gpu_test.py)\n")
    os.makedirs(os.path.join(OUT, "rmm"), exist_ok=True)
    with open(os.path.join(OUT, "rmm", "__init__.py"), "w") as f:
        f.write('"""Stub: device memory is plain hipMalloc inside libimplicit_hip.so."""\n')
    os.makedirs(os.path.join(OUT, "tests"), exist_ok=True)
    for t in TESTS:
        shutil.copy(os.path.join(reference, "tests", t), os.path.join(OUT, "tests"))
    if verbose:
        print(f"[oracle/refsuite] assembled {OUT}")
    return True


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(0 if assemble() else 1)
