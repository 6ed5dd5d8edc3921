import os
import sys
import warnings

# scipy-openblas is built for <= 64 threads and crashes on the 256-core GPU box without this
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import implicit_amd.gpu as g
    return g.HAS_CUDA


@pytest.fixture(scope="session")
def gpu():
    """The implicit_amd.gpu module; GPU tests FAIL (not skip) when the HIP library or device is
    missing -- there is no fallback path to silently pass on."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import implicit_amd.gpu as g
    assert g.HAS_CUDA, "libimplicit_hip.so not loadable or no HIP device: GPU tests cannot run"
    return g


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o
