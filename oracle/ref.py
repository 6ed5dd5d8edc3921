"""Loader for the compiled reference modules in oracle/_ref (TEST INFRASTRUCTURE ONLY).

`load()` returns (als, topk): the reference's own `implicit.cpu._als` and
`implicit.cpu.topk` extension modules, built by oracle/build_ref.py from
/root/reference/implicit/cpu/{_als.pyx,topk.pyx,select.h}.  Returns (None, None)
when they have not been built.  Never imported by the product package.
"""
import importlib.util
import os
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def _load(name):
    if name in _cache:
        return _cache[name]
    path = os.path.join(_HERE, "_ref", name + sysconfig.get_config_var("EXT_SUFFIX"))
    mod = None
    if os.path.exists(path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _cache[name] = mod
    return mod


def load():
    return _load("_als"), _load("topk")


def available():
    a, t = load()
    return a is not None and t is not None
