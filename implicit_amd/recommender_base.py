"""API contract of every model in this package (implicit/recommender_base.py:13-223)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class ModelFitError(Exception):
    """NaN factors after fit (implicit/recommender_base.py:9, 218-223)."""


class RecommenderBase(metaclass=ABCMeta):
    @abstractmethod
    def fit(self, user_items, show_progress=True, callback=None):
        """Train on a (users x items) CSR matrix of confidences."""

    @abstractmethod
    def recommend(self, userid, user_items, N=10, filter_already_liked_items=True, filter_items=None,
                  recalculate_user=False, items=None):
        """Top-N items for a user or a batch of users -> (ids, scores)."""

    @abstractmethod
    def similar_users(self, userid, N=10, filter_users=None, users=None):
        """Top-N users by cosine similarity -> (ids, scores)."""

    @abstractmethod
    def similar_items(self, itemid, N=10, recalculate_item=False, item_users=None, filter_items=None, items=None):
        """Top-N items by cosine similarity -> (ids, scores)."""

    @abstractmethod
    def save(self, file):
        """numpy .npz checkpoint."""

    @classmethod
    def load(cls, fileobj_or_path):
        """Inverse of save(): every npz entry becomes an attribute (recommender_base.py:173-202)."""
        if isinstance(fileobj_or_path, str) and not fileobj_or_path.endswith(".npz"):
            fileobj_or_path += ".npz"
        with np.load(fileobj_or_path, allow_pickle=False) as data:
            model = cls()
            for key, value in data.items():
                if key == "dtype":
                    value = np.dtype(str(value))
                elif value.shape == ():
                    value = value.item()
                setattr(model, key, value)
        return model

    @staticmethod
    def _check_factors(user_factors, item_factors):
        if np.isnan(user_factors).any() or np.isnan(item_factors).any():
            raise ModelFitError("NaN encountered in factors")
