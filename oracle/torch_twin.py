"""Independent torch-CPU fp32 twin of the Keras forward + the timed CPU baseline.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two uses:
  1. cross-check of `oracle/ref_np.py` by a second, independently written
     implementation (torch's conv1d / linear instead of explicit tap loops), since
     TensorFlow -- the reference's actual backend -- cannot be installed here;
  2. `bench.py`'s `cpu_baseline` leg: the reference-style CPU path of
     keras_model.py:69-79 -- per-character Python encode loop
     (sequence_utils.py:44-47), float32 tensor, forward in 256-row batches
     (`predict(batch_size=256)`, keras_model.py:78) on all host cores,
     `np.stack` / `np.mean` for the ensemble (ensemble.py:55-59).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_np


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _conv_same(h, w, b):
    # h (N, C, L); Keras 'same': pad_left=(k-1)//2, extra zero on the RIGHT
    k = w.shape[0]
    pl = (k - 1) // 2
    pr = k - 1 - pl
    h = F.pad(h, (pl, pr))
    return F.conv1d(h, _t(w).permute(2, 1, 0).contiguous(), _t(b))


def cnn_forward(x, weights):
    """cnn.py:23-54; x float32 (N, L, A) torch tensor -> (N,) float32."""
    w1, b1, w2, b2, w3, b3, d1, c1, d2, c2, d3, c3 = weights
    h = x.permute(0, 2, 1)                                   # channels-first
    if h.shape[2] < w1.shape[0]:
        raise ValueError("valid conv with L < kernel_size")
    h = F.relu(F.conv1d(h, _t(w1).permute(2, 1, 0).contiguous(), _t(b1)))
    h = F.relu(_conv_same(h, w2, b2))
    h = F.relu(_conv_same(h, w3, b3))
    h = h.amax(dim=2)
    h = F.relu(F.linear(h, _t(d1).T, _t(c1)))
    h = F.relu(F.linear(h, _t(d2).T, _t(c2)))
    return F.linear(h, _t(d3).T, _t(c3))[:, 0]


def mlp_forward(x, weights):
    d1, c1, d2, c2, d3, c3, d4, c4 = weights
    h = x.reshape(x.shape[0], -1)
    h = F.relu(F.linear(h, _t(d1).T, _t(c1)))
    h = F.relu(F.linear(h, _t(d2).T, _t(c2)))
    h = F.relu(F.linear(h, _t(d3).T, _t(c3)))
    return F.linear(h, _t(d4).T, _t(c4))[:, 0]


FORWARD = {"cnn": cnn_forward, "mlp": mlp_forward, "ge": mlp_forward}


def predict(one_hots_f32: np.ndarray, kind: str, weights, batch_size: int = 256):
    """`model.predict(one_hots, batch_size=256)` stand-in (keras_model.py:78)."""
    x = torch.from_numpy(one_hots_f32)
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch_size):
            outs.append(FORWARD[kind](x[i:i + batch_size], weights))
    y = torch.cat(outs).numpy() if outs else np.zeros((0,), np.float32)
    return np.nan_to_num(y)


def keras_fitness_cpu(sequences, alphabet, kind, weights, batch_size=256, loop_encode=True):
    """Reference-style CPU path for ONE KerasModel.get_fitness call."""
    if loop_encode:
        x = ref_np.encode_batch_loop(sequences, alphabet)
    else:
        x = ref_np.encode_batch(sequences, alphabet).astype(np.float32)
    return predict(x, kind, weights, batch_size)


def ensemble_fitness_cpu(sequences, alphabet, kind, weight_sets, batch_size=256, loop_encode=True):
    """ensemble.py:54-59: M sequential member calls (each re-encodes), stack, mean."""
    scores = np.stack(
        [keras_fitness_cpu(sequences, alphabet, kind, w, batch_size, loop_encode) for w in weight_sets],
        axis=1,
    )
    return np.mean(scores, axis=1)
