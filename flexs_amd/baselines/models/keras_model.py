"""`KerasModel` -- the wrapper the reference puts around a tf.keras model
(flexs/baselines/models/keras_model.py:12-79) -- re-built around the MI355X
scoring engine.

What was a `tf.keras.Sequential` is an `Architecture` here: a shape description
plus the weight arrays in Keras `get_weights()` order.  `get_fitness` runs the
fused encode + forward HIP kernels (libflexs_amd.so, fx_score); `train` is the
PyTorch-ROCm replacement of `model.fit` (Adam / MSE / 20 epochs / batch 256,
the Keras defaults the reference compiles with) followed by a weight upload to
the engine -- weights change once per explorer round (flexs/explorer.py:157).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

import flexs_amd
from flexs_amd import _native
from flexs_amd.types import SEQUENCES_TYPE

KINDS = {"cnn": _native.FX_CNN, "mlp": _native.FX_MLP, "ge": _native.FX_GE}


class Architecture:
    """Stand-in for the compiled tf.keras model object held in `KerasModel.model`."""

    def __init__(self, kind: str, seq_len: int, alphabet_size: int, hidden_size: int, num_filters: int = 0,
                 kernel_size: int = 0, loss: str = "MSE", seed: Optional[int] = None):
        if kind not in KINDS:
            raise ValueError(f"unknown architecture {kind!r}")
        self.kind = kind
        self.L, self.A, self.H, self.F, self.K = seq_len, alphabet_size, hidden_size, num_filters, kernel_size
        self.loss = loss
        if kind == "cnn" and seq_len < kernel_size:
            # Keras raises while building Conv1D(padding="valid") (cnn.py:25-32)
            raise ValueError(
                f"Negative dimension size caused by subtracting {kernel_size} from {seq_len} for 'valid' Conv1D"
            )
        self._weights = self._initial_weights(np.random.default_rng(seed))

    # -- shapes in Keras get_weights() order (cnn.py:23-54, mlp.py:21-31, global_epistasis_model.py:26-36)
    def shapes(self):
        L, A, H, F, K = self.L, self.A, self.H, self.F, self.K
        if self.kind == "cnn":
            return [(K, A, F), (F,), (K, F, F), (F,), (A - 1, F, F), (F,), (F, H), (H,), (H, H), (H,), (H, 1), (1,)]
        if self.kind == "mlp":
            return [(L * A, H), (H,), (H, H), (H,), (H, H), (H,), (H, 1), (1,)]
        return [(L * A, 1), (1,), (1, H), (H,), (H, H), (H,), (H, 1), (1,)]

    def _initial_weights(self, rng):
        """Keras defaults: glorot_uniform kernels, zero biases."""
        out = []
        for shp in self.shapes():
            if len(shp) == 1:
                out.append(np.zeros(shp, np.float32))
            else:
                receptive = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
                lim = np.sqrt(6.0 / (shp[-2] * receptive + shp[-1] * receptive))
                out.append(rng.uniform(-lim, lim, shp).astype(np.float32))
        return out

    def get_weights(self) -> List[np.ndarray]:
        return [w.copy() for w in self._weights]

    def set_weights(self, weights):
        shapes = self.shapes()
        if len(weights) != len(shapes):
            raise ValueError(f"expected {len(shapes)} weight arrays, got {len(weights)}")
        new = []
        for w, shp in zip(weights, shapes):
            w = np.asarray(w, np.float32)
            if w.shape != tuple(shp):
                raise ValueError(f"weight shape {w.shape} does not match {shp}")
            new.append(np.ascontiguousarray(w))
        self._weights = new
        self._version = getattr(self, "_version", 0) + 1

    def count_params(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.shapes()))


class KerasModel(flexs_amd.Model):
    """Same constructor and methods as the reference wrapper (keras_model.py:15-47);
    `model` is an `Architecture` instead of a tf.keras model."""

    def __init__(
        self,
        model: Architecture,
        alphabet: str,
        name: str,
        batch_size: int = 256,
        epochs: int = 20,
        custom_train_function: Callable = None,
        custom_predict_function: Callable = None,
        device: Optional[int] = None,
    ):
        super().__init__(name)
        self.model = model
        self.alphabet = alphabet
        self.name = name
        self.epochs = epochs
        self.batch_size = batch_size
        self._device = device
        self._lut = _native.make_lut(alphabet)
        self._native_model = None
        self._native_version = None
        self._small = None

    # ------------------------------------------------------------------ copy / pickle: device handles stay behind
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_native_model"] = None          # the copy uploads its weights to its own fx_model on first use
        state["_native_version"] = None
        state["_small"] = None
        return state

    # ------------------------------------------------------------------ engine plumbing
    @property
    def seq_len(self) -> int:
        return self.model.L

    def _engine(self):
        return _native.Engine.get(self._device)

    def native(self):
        """fx_model handle with the current weights (created / refreshed lazily)."""
        a = self.model
        if self._native_model is None:
            self._native_model = _native.NativeModel(self._engine(), KINDS[a.kind], a.L, a.A, a.F, a.H, a.K)
            self._native_version = None
        version = (id(a), getattr(a, "_version", 0))
        if self._native_version != version:
            self._native_model.set_weights(a._weights)
            self._native_version = version
        return self._native_model

    # ------------------------------------------------------------------ flexs.Model API
    def train(self, sequences: SEQUENCES_TYPE, labels: np.ndarray, verbose: bool = False, seed: Optional[int] = None):
        """Replacement of `self.model.fit(one_hots, labels, batch_size, epochs)`
        (keras_model.py:49-67) in PyTorch; see flexs_amd/training.py.  `seed` (not in the reference) fixes the
        shuffles and dropout masks of this call."""
        from flexs_amd import training

        training.fit(self.model, sequences, labels, self.alphabet, batch_size=self.batch_size,
                     epochs=self.epochs, verbose=verbose, seed=seed)

    def _fitness_function(self, sequences):
        """keras_model.py:69-79: encode -> float32 tensor -> predict -> squeeze ->
        nan_to_num, fused on the GPU.  Returns float32 (N,)."""
        if (type(sequences) is np.ndarray and sequences.dtype.kind == "U" and sequences.ndim == 1
                and 0 < sequences.shape[0] <= _native.SMALL_CALL_ROWS):
            sequences = sequences.tolist()                      # (explorers also pass small NumPy string arrays: same fast path)
        if type(sequences) in (list, tuple) and 0 < len(sequences) <= _native.SMALL_CALL_ROWS and _native._HAS_SCORE_SMALL:
            # explorer-size call: string packing + fx_score in one C call on a cached argument block
            nat = self.native()
            c = self._small
            if c is None or c[0] is not nat:
                c = self._small = (nat, _native.small_plan(self._engine(), [nat], self.model.L, self._lut, False) or b"")
            if c[1]:
                out = _native.score_small(self._engine(), c[1], sequences, 1, False)    # (N, 1)
                if out is not None:
                    return out.reshape(len(sequences))
        if _native.wants_chunked(sequences, self.model.L):
            nm, _ = self._engine().score_strings([self.native()], sequences, self.model.L, self._lut, want_matrix=True)
            return nm[:, 0]
        seq_bytes = _native.sequences_to_bytes(sequences, L=self.model.L, staging=self._engine())
        if seq_bytes.shape[0] == 0:
            return np.zeros((0,), np.float32)
        nm, _ = self._engine().score([self.native()], seq_bytes, self._lut, want_matrix=True)
        return nm[:, 0]
