"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/implicit_hip.h declares, maps errors as documented, and the Python shim keeps the
reference's import contract (HAS_CUDA False + warning on a box without a device)."""
import ctypes
import os
import re
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "implicit_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(imp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from implicit_amd import _build
    from implicit_amd.gpu import _hip

    if not os.path.exists(_hip.LIB_PATH):
        _build.build(verbose=False)
    return _hip.lib()


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 45
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/implicit_hip.h but not exported"


def test_shim_binds_exactly_the_header(lib):
    from implicit_amd.gpu import _hip

    assert sorted(_hip.EXPORTED_SYMBOLS) == declared_functions()


def test_no_torch_in_library_dependencies(lib):
    from implicit_amd.gpu import _hip

    import subprocess

    dynamic = subprocess.run(["readelf", "-d", _hip.LIB_PATH], capture_output=True, text=True).stdout
    needed = "\n".join(line for line in dynamic.splitlines() if "(NEEDED)" in line)  # library names only, not addresses
    assert "torch" not in needed and "c10" not in needed
    assert "librccl" in needed and "libamdhip64" in needed


def test_error_mapping_without_device(lib):
    from implicit_amd.gpu import _hip

    has_device = ctypes.c_int(0)
    status = lib.imp_get_device_count(ctypes.byref(has_device))
    if status == _hip.IMP_OK:
        pytest.skip("a device is present: covered by the gpu suite")
    assert status == _hip.IMP_RUNTIME_ERROR
    assert b"HIP error" in lib.imp_last_error() or b"no HIP device" in lib.imp_last_error()
    with pytest.raises(RuntimeError):
        _hip.check(status)


def test_import_contract_without_gpu():
    """implicit/gpu/__init__.py:8-30: import succeeds, HAS_CUDA False, constructing the model raises."""
    import importlib

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        import implicit_amd.gpu as g

        g = importlib.reload(g)
    if g.HAS_CUDA:
        pytest.skip("a device is present")
    assert any("disabling GPU support" in str(x.message) or "Disabling GPU support" in str(x.message) for x in w)
    import implicit_amd.gpu.als as als

    with pytest.raises(ValueError):
        als.AlternatingLeastSquares(factors=8)
    from implicit_amd.als import AlternatingLeastSquares

    with pytest.raises(ValueError):
        AlternatingLeastSquares(factors=8, use_gpu=False)


def test_product_never_imports_the_oracle():
    """A product path that routes through oracle/ voids every parity claim."""
    pkg = os.path.join(ROOT, "implicit_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "liboracle" not in text and "oracle/_ref" not in text, fn
