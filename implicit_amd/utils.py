"""Small host-side helpers shared by the model layer (semantics of implicit/utils.py)."""
import time
import warnings

import numpy as np
import scipy.sparse


class ParameterWarning(Warning):
    """Raised (as a warning) when an input had to be converted (implicit/utils.py:155-156)."""


def check_csr(user_items):
    """Accepts only CSR; anything else is converted with a ParameterWarning (implicit/utils.py:159-169)."""
    if isinstance(user_items, scipy.sparse.csr_matrix):
        return user_items
    kind = type(user_items).__name__
    t0 = time.time()
    converted = user_items.tocsr()
    warnings.warn(
        f"Method expects CSR input, and was passed {kind} instead. "
        f"Converting to CSR took {time.time() - t0} seconds",
        ParameterWarning,
    )
    return converted


def check_random_state(random_state):
    """numpy Generator from None / int / RandomState / Generator (implicit/utils.py:65-83; the
    reference's RandomState branch calls a non-existent `rand_int` -- here it works)."""
    if isinstance(random_state, np.random.RandomState):
        return np.random.default_rng(random_state.randint(2**31))
    return np.random.default_rng(random_state)


def nonzeros(m, row):
    """(index, value) pairs of one CSR row (implicit/utils.py:9-12)."""
    for k in range(m.indptr[row], m.indptr[row + 1]):
        yield m.indices[k], m.data[k]
