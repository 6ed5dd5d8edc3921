"""`implicit.gpu` for MI355X: same import contract as implicit/gpu/__init__.py:5-30.

Importing succeeds on a box without a GPU (or without the built library): HAS_CUDA is then False
and a warning says why.  HAS_RMM is kept for API compatibility (the reference's model constructor
checks it, implicit/gpu/als.py:58-62); there is no RMM here -- device memory is plain hipMalloc.
"""
import warnings

HAS_CUDA = False
HAS_RMM = False

try:
    from ._cuda import (COOMatrix, Comm, CSRMatrix, IntVector, KnnQuery, LeastSquaresSolver, Matrix,  # noqa: F401
                        Profiler, RandomState, bpr_update, calculate_norms, core_clock_mhz, debug_occupy, fixup_rows, get_device, get_device_count, get_oversubscribe, release_workspaces,
                        set_deferred_sync, set_device, set_oversubscribe, synchronize)
    from ._hip import lib as _lib

    _lib()  # ImportError when libimplicit_hip.so has not been built
    HAS_RMM = True
    get_device_count()  # RuntimeError when no device is usable
    HAS_CUDA = True
except RuntimeError as e:  # library present, no usable device
    warnings.warn(f"HIP extension is built, but disabling GPU support because of '{e}'")
except (ImportError, OSError) as e:
    warnings.warn(f"Disabling GPU support because of '{e}'")
