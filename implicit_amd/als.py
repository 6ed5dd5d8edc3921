"""Factory with the signature of implicit/als.py:7-80.  Only the GPU branch exists in this package:
the reference's CPU model is the parity oracle (oracle/), not part of the product."""
import numpy as np

import implicit_amd.gpu


def AlternatingLeastSquares(factors=100, regularization=0.01, alpha=1.0, dtype=np.float32, use_native=True,
                            use_cg=True, use_gpu=None, iterations=15, calculate_training_loss=False,
                            num_threads=0, random_state=None, **gpu_kwargs):
    """`gpu_kwargs`: keyword arguments of the MI355X model that the reference's factory does not have (`comm=` for the
    multi-GPU fit, `cg_steps=`, `init=`), passed through."""
    if use_gpu is None:
        use_gpu = implicit_amd.gpu.HAS_CUDA
    if not use_gpu:
        raise ValueError("implicit_amd only ships the MI355X (use_gpu=True) path; "
                         "use benfred/implicit for the CPU model")
    import implicit_amd.gpu.als

    return implicit_amd.gpu.als.AlternatingLeastSquares(
        factors, regularization, alpha, dtype=dtype, iterations=iterations,
        calculate_training_loss=calculate_training_loss, random_state=random_state, use_cg=use_cg, **gpu_kwargs)
