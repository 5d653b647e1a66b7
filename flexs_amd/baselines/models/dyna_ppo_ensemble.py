"""`DynaPPOEnsemble` -- the r^2-gated ensemble of
flexs/baselines/explorers/dyna_ppo.py:32-130, with device-backed members.

The reference's default member list is 3 Keras surrogates + 8 scikit-learn
regressors (dyna_ppo.py:51-86); the scikit-learn wrappers are third-party CPU
estimators outside the rebuilt hot path, so the default
here is the three Keras-type members only -- pass `models=` to add others
(any object with the flexs.Model interface works: members are only ever
touched through `train` / `get_fitness`).
"""
from typing import List, Optional

import numpy as np
import scipy.stats
import sklearn.model_selection

import flexs_amd
from flexs_amd import _native
from flexs_amd.ensemble import _device_members, train_members


class DynaPPOEnsemble(flexs_amd.Model):
    def __init__(self, seq_len: int, alphabet: str, r_squared_threshold: float = 0.5,
                 models: Optional[List[flexs_amd.Model]] = None):
        super().__init__(name="DynaPPOEnsemble")                       # dyna_ppo.py:48
        if models is None:
            from flexs_amd.baselines.models import CNN, MLP, GlobalEpistasisModel

            models = [                                                 # dyna_ppo.py:53-55
                GlobalEpistasisModel(seq_len, 100, alphabet),
                MLP(seq_len, 200, alphabet),
                CNN(seq_len, 32, 100, alphabet),
            ]
        self.models = models
        self.r_squared_vals = np.ones(len(self.models))
        self.r_squared_threshold = r_squared_threshold

    def train(self, sequences, labels):
        """Hold out 25 %, train every member, score r^2 on the hold-out (dyna_ppo.py:92-116)."""
        if len(sequences) < 10:
            return
        (train_X, test_X, train_y, test_y) = sklearn.model_selection.train_test_split(
            np.array(sequences), np.array(labels), test_size=0.25
        )
        train_members(self.models, train_X, train_y)
        self.r_squared_vals = []
        for model in self.models:
            y_preds = model.get_fitness(test_X)
            if (y_preds[0] == y_preds).all() or (test_y[0] == test_y).all():
                self.r_squared_vals.append(0)                          # constant -> r^2 undefined -> 0
            else:
                self.r_squared_vals.append(scipy.stats.pearsonr(test_y, model.get_fitness(test_X))[0] ** 2)

    def _fitness_function(self, sequences):
        passing = [m for m, r2 in zip(self.models, self.r_squared_vals) if r2 >= self.r_squared_threshold]
        if len(passing) == 0:                                          # dyna_ppo.py:125-126
            return self.models[int(np.argmax(self.r_squared_vals))].get_fitness(sequences)
        if _device_members(passing) and len(sequences):
            # one fused launch for all passing members (each is charged like a get_fitness call)
            for m in passing:
                m.cost += len(sequences)
            m0 = passing[0]
            seq_bytes = _native.sequences_to_bytes(sequences, L=m0.model.L)
            nm, _ = m0._engine().score([m.native() for m in passing], seq_bytes, m0._lut, want_matrix=True)
            cols = [np.ascontiguousarray(nm[:, j]) for j in range(len(passing))]
        else:
            cols = [m.get_fitness(sequences) for m in passing]
        return np.mean(cols, axis=0)                                   # dyna_ppo.py:128-130 (member axis first)
