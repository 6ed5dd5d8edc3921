"""Inference side of the GPU matrix-factorisation models: recommend / similar_items / similar_users
over KnnQuery.topk.  Same public behaviour as implicit/gpu/matrix_factorization_base.py:12-259
(which in turn follows implicit/cpu/matrix_factorization_base.py:35-264)."""
import time

import numpy as np
from scipy.sparse import csr_matrix

import implicit_amd.gpu as gpu

from ..recommender_base import RecommenderBase


def _filter_items_from_sparse_matrix(items, query_items):
    """Re-index the liked-items matrix onto positions inside the sorted `items` subset, dropping
    entries that are not in the subset (cpu/matrix_factorization_base.py:253-264)."""
    coo = query_items.tocoo()
    pos = np.clip(np.searchsorted(items, coo.col), 0, len(items) - 1)
    keep = items[pos] == coo.col
    return csr_matrix((coo.data[keep], (coo.row[keep], pos[keep])), shape=(coo.shape[0], len(items)))


class MatrixFactorizationBase(RecommenderBase):
    """item_factors / user_factors are implicit_amd.gpu.Matrix objects resident in HBM."""

    def __init__(self):
        self.item_factors = None
        self.user_factors = None
        self._item_norms = self._user_norms = None
        self._item_norms_host = self._user_norms_host = None
        self._knn = None

    # ---- recommend ---------------------------------------------------------------------------------
    def recommend(self, userid, user_items, N=10, filter_already_liked_items=True, filter_items=None,
                  recalculate_user=False, items=None):
        scalar = np.isscalar(userid)
        if filter_already_liked_items or recalculate_user:
            if not isinstance(user_items, csr_matrix):
                raise ValueError("user_items needs to be a CSR sparse matrix")
            if user_items.shape[0] != (1 if scalar else len(userid)):
                raise ValueError("user_items must contain 1 row for every user in userids")

        if recalculate_user:
            query = self.recalculate_user(userid, user_items)
        elif not scalar and len(userid) > 1 and isinstance(userid, np.ndarray) and userid.dtype.kind in "iu" and \
                userid[0] >= 0 and userid[-1] - userid[0] == len(userid) - 1 and (np.diff(userid) == 1).all():
            # a run of consecutive ids (what batched callers pass): a row-range VIEW of the factors, no gather
            if userid[-1] >= self.user_factors.shape[0]:
                raise IndexError("row id out of range for selecting items from matrix")
            query = self.user_factors[int(userid[0]):int(userid[-1]) + 1]
        else:
            query = self.user_factors[userid]

        candidates = self.item_factors
        if items is not None:
            if filter_items:
                raise ValueError("Can't set both items and filter_items in recommend call")
            N = min(N, len(items))
            items = np.sort(np.array(items))
            if items.max() >= self.item_factors.shape[0] or items.min() < 0:
                raise IndexError("Some itemids are not in the model")
            candidates = candidates[items]

        item_filter = None
        if filter_items is not None:
            item_filter = gpu.IntVector(np.array(filter_items, dtype="int32"))

        query_filter = None
        if filter_already_liked_items:
            liked = user_items if items is None else _filter_items_from_sparse_matrix(items, user_items)
            if liked.nnz:
                query_filter = gpu.COOMatrix.from_csr_pattern(liked)  # the filter reads (row, col) only

        ids, scores = self.knn.topk(candidates, query, N, query_filter=query_filter, item_filter=item_filter)
        if scalar:
            ids, scores = ids[0], scores[0]
        if items is not None:
            ids = items[ids]
        return ids, scores

    # ---- cosine similarity -------------------------------------------------------------------------
    def _similar(self, factors, norms_dev, norms_host, queryid, N, subset, filter_ids, what, query_factors=None):
        candidates, cand_norms = factors, norms_dev
        if subset is not None:
            if filter_ids:
                raise ValueError(f"Can't set both {what} and filter_{what} in similar_{what} call")
            subset = np.array(subset)
            if subset.max() >= factors.shape[0] or subset.min() < 0:
                raise IndexError(f"Some ids in the {what} parameter are not in the model")
            candidates = factors[subset]
            cand_norms = gpu.Matrix(norms_host[subset].reshape(1, len(subset)))
        item_filter = None
        if filter_ids is not None:
            item_filter = gpu.IntVector(np.array(filter_ids, dtype="int32"))
        if query_factors is None:
            query_factors = factors[queryid]
        ids, scores = self.knn.topk(candidates, query_factors, N, cand_norms, item_filter=item_filter)
        if subset is not None:
            ids = subset[ids]
        qnorm = norms_host[queryid]
        if np.isscalar(queryid):
            ids, scores = ids[0], scores[0]
            scores /= qnorm
        else:
            scores /= qnorm[:, None]
        return ids, scores

    def similar_users(self, userid, N=10, filter_users=None, users=None):
        norms = self.user_norms
        return self._similar(self.user_factors, norms, self._user_norms_host, userid, N, users, filter_users, "users")

    def similar_items(self, itemid, N=10, recalculate_item=False, item_users=None, filter_items=None, items=None):
        norms = self.item_norms
        query = self.recalculate_item(itemid, item_users) if recalculate_item else None
        return self._similar(self.item_factors, norms, self._item_norms_host, itemid, N, items, filter_items, "items",
                             query_factors=query)

    @property
    def user_norms(self):
        if self._user_norms is None:
            self._user_norms = gpu.calculate_norms(self.user_factors)
            self._user_norms_host = self._user_norms.to_numpy().reshape(self._user_norms.shape[1])
        return self._user_norms

    @property
    def item_norms(self):
        if self._item_norms is None:
            self._item_norms = gpu.calculate_norms(self.item_factors)
            self._item_norms_host = self._item_norms.to_numpy().reshape(self._item_norms.shape[1])
        return self._item_norms

    @property
    def knn(self):
        if self._knn is None:
            self._knn = gpu.KnnQuery()
        return self._knn

    def recalculate_user(self, userid, user_items):
        raise NotImplementedError("recalculate_user is not supported with this model")

    def recalculate_item(self, itemid, item_users):
        raise NotImplementedError("recalculate_item is not supported with this model")

    def _check_fit_errors(self):
        """NaN factors raise ModelFitError (cpu/matrix_factorization_base.py:249-250).  Checked through the row norms,
        computed on the device: a NaN anywhere in a row makes that row's norm NaN, and rows x 4 bytes come back instead
        of both factor matrices (0.33 GB at configs[2] -- a fifth of fit()'s set-up time)."""
        self._check_factors(gpu.calculate_norms(self.user_factors).to_numpy(), gpu.calculate_norms(self.item_factors).to_numpy())

    # ---- pickling: device arrays travel as numpy (gpu/matrix_factorization_base.py:220-234) --------
    def __getstate__(self):
        state = self.__dict__.copy()
        for attr in ("_knn", "_user_norms", "_user_norms_host", "_item_norms", "_item_norms_host"):
            state[attr] = None
        state["item_factors"] = self.item_factors.to_numpy() if self.item_factors else None
        state["user_factors"] = self.user_factors.to_numpy() if self.user_factors else None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        if self.item_factors is not None:
            self.item_factors = gpu.Matrix(self.item_factors)
        if self.user_factors is not None:
            self.user_factors = gpu.Matrix(self.user_factors)


def check_random_state(random_state):
    """A device RandomState from None / int / numpy RandomState / Generator
    (gpu/matrix_factorization_base.py:237-259)."""
    if isinstance(random_state, np.random.RandomState):
        return gpu.RandomState(random_state.randint(2**31))
    if isinstance(random_state, np.random.Generator):
        return gpu.RandomState(random_state.integers(2**31))
    return gpu.RandomState(random_state or int(time.time()))
