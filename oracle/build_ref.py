#!/usr/bin/env python3
"""Build the *real* reference CPU path (benfred/implicit's Cython ALS solvers and
top-k selector) from the sources where they lie under /root/reference.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.

Recipe: `cython --cplus` on implicit/cpu/_als.pyx and
implicit/cpu/topk.pyx (the latter includes implicit/cpu/select.h) followed by
g++ -O3 -fopenmp.  The reference's own cmake/scikit-build is NOT run.  All
outputs (generated .cxx and the extension modules) go to oracle/_ref/, which is
git-ignored (never committed: no reference source enters the history) but is
shipped to the GPU box next to our own built .so files.

Usage:  python oracle/build_ref.py [--reference /root/reference] [--force]
Exit status 0 and a no-op when the reference tree is absent (GPU box).
"""
import argparse
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
MODULES = {
    "_als": "implicit/cpu/_als.pyx",
    "topk": "implicit/cpu/topk.pyx",
}
# only needed for the reference PACKAGE to import when its own tests are run over the shim (oracle/refsuite.py): the
# reference's __init__ and test mixin pull in these out-of-scope models
EXTRA_MODULES = {
    "bpr": "implicit/cpu/bpr.pyx",
    "lmf": "implicit/cpu/lmf.pyx",
    "evaluation": "implicit/evaluation.pyx",
    "_nearest_neighbours": "implicit/_nearest_neighbours.pyx",
}
# directives copied from the reference's implicit/CMakeLists.txt:1-5
DIRECTIVES = "always_allow_keywords=True,binding=True,embedsignature=True,language_level=3"


def ext_path(name):
    return os.path.join(OUT, name + sysconfig.get_config_var("EXT_SUFFIX"))


def build(reference="/root/reference", force=False, verbose=True, extra=False):
    modules = dict(MODULES, **EXTRA_MODULES) if extra else MODULES
    if not os.path.isdir(reference):
        if verbose:
            print(f"[oracle/_ref] {reference} absent: keeping prebuilt files (if any)")
        return all(os.path.exists(ext_path(m)) for m in modules)
    import numpy

    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, rel in modules.items():
        src = os.path.join(reference, rel)
        so = ext_path(name)
        if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
            continue
        cxx = os.path.join(OUT, name + ".cxx")
        subprocess.check_call(
            ["cython", "--cplus", "-3", "--directive", DIRECTIVES, "-I", reference, src, "-o", cxx]
        )
        cmd = [
            "g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c++17", "-w",
            "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(), "-I", reference,
            cxx, "-o", so,
        ]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"g++ failed for reference module {name}")
        cxx = os.path.join(OUT, name + ".cxx")
        if os.path.exists(cxx):  # generated from reference source: keep only the binary
            os.remove(cxx)
        if verbose:
            print(f"[oracle/_ref] built {ext_path(name)}")
    return True


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    sys.exit(0 if build(a.reference, a.force) else 1)
