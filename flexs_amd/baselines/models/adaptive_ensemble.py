"""`AdaptiveEnsemble` -- same contract as flexs/baselines/models/adaptive_ensemble.py:29-102."""
from typing import List

import numpy as np
import scipy.stats
import sklearn.model_selection

import flexs_amd
from flexs_amd import _native
from flexs_amd.ensemble import _device_members
from flexs_amd.types import SEQUENCES_TYPE


def r2_weights(model_preds: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """Normalised squared Pearson r per model (adaptive_ensemble.py:12-26)."""
    r2s = np.array([scipy.stats.pearsonr(preds, labels)[0] ** 2 for preds in model_preds])
    return r2s / r2s.sum()


def _weighted_sum(w, x):
    return np.sum(w * x, axis=1)           # adaptive_ensemble.py:54


class AdaptiveEnsemble(flexs_amd.Model):
    """Ensemble whose members are re-weighted (r^2 on a validation split) at every `train`."""

    def __init__(
        self,
        models: List[flexs_amd.Model],
        combine_with="sum",
        adapt_weights_with="r2_weights",
        adaptive_val_size: float = 0.2,
    ):
        name = f"AdaptiveEns({'|'.join(model.name for model in models)})"
        super().__init__(name)
        self.models = models
        self.weights = np.ones(len(models)) / len(models)
        if combine_with == "sum":
            combine_with = _weighted_sum
        self.combine_with = combine_with
        if adapt_weights_with == "r2_weights":
            adapt_weights_with = r2_weights
        self.adapt_weights_with = adapt_weights_with
        self.adaptive_val_size = adaptive_val_size

    def train(self, sequences: SEQUENCES_TYPE, labels: np.ndarray):
        if len(sequences) < 10:                                     # adaptive_ensemble.py:82-85
            for model in self.models:
                model.train(sequences, labels)
            return
        (train_X, test_X, train_y, test_y) = sklearn.model_selection.train_test_split(
            np.array(sequences), np.array(labels), test_size=self.adaptive_val_size
        )
        for model in self.models:
            model.train(train_X, train_y)
        preds = np.stack([model.get_fitness(test_X) for model in self.models], axis=0)
        self.weights = self.adapt_weights_with(preds, test_y)

    def _fitness_function(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
        if _device_members(self.models) and self.combine_with is _weighted_sum and len(sequences):
            n = len(sequences)
            for m in self.models:
                m.cost += n                                          # members are scored through get_fitness (:98-100)
            m0 = self.models[0]
            seq_bytes = _native.sequences_to_bytes(sequences, L=m0.model.L)
            nm, _ = m0._engine().score([m.native() for m in self.models], seq_bytes, m0._lut, want_matrix=True)
            return m0._engine().ensemble_weighted_sum(nm, np.asarray(self.weights, np.float64))
        scores = np.stack([model.get_fitness(sequences) for model in self.models], axis=1)
        return self.combine_with(self.weights, scores)
