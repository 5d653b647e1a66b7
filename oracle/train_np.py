"""float64 NumPy restatement of ONE Keras training step of the reference surrogates.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates what
`model.compile(loss="MSE", optimizer="adam")` + one `train_on_batch` do for the
three architectures (flexs/baselines/models/cnn.py:23-56, mlp.py:21-33,
global_epistasis_model.py:26-37, driven by keras_model.py:60-67):

* forward in training mode (Dropout(0.25) before the CNN's last Dense, cnn.py:51;
  Keras scales the kept units by 1 / (1 - rate));
* loss = mean over the batch of (prediction - label)^2 (Keras "MSE": mean over the
  last axis, then over the batch; labels (n,) are expanded to (n, 1));
* reverse-mode gradients written out by hand (dense, ReLU with gradient 0 at 0,
  GlobalMaxPooling1D routing the gradient to the maxima -- evenly on ties, as
  TensorFlow's reduce_max gradient does -- Conv1D valid / same);
* tf.keras.optimizers.Adam defaults (learning_rate 1e-3, beta_1 .9, beta_2 .999,
  epsilon 1e-7, amsgrad False) in Keras' formulation:
      t <- t + 1;  lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)
      m <- beta_1 m + (1 - beta_1) g;  v <- beta_2 v + (1 - beta_2) g^2
      w <- w - lr_t * m / (sqrt(v) + epsilon)
  with (t, m, v) persisting across calls (the optimiser belongs to the compiled model).

TensorFlow cannot be installed here: like the forward, this is
restated from Keras' documented behaviour -> training parity is UNPINNED; what the
tests pin is that flexs_amd/training.py does exactly this arithmetic.
"""
from __future__ import annotations

import numpy as np

LR, BETA_1, BETA_2, EPSILON, DROPOUT = 1e-3, 0.9, 0.999, 1e-7, 0.25


def _pad_same(x, k):
    pl = (k - 1) // 2
    return np.pad(x, ((0, 0), (pl, k - 1 - pl), (0, 0))), pl


def _conv_fwd(x, w, b, same):
    k = w.shape[0]
    xp = _pad_same(x, k)[0] if same else x
    lout = xp.shape[1] - k + 1
    out = np.broadcast_to(b, (x.shape[0], lout, w.shape[2])).copy()
    for j in range(k):
        out += xp[:, j:j + lout, :] @ w[j]
    return out, xp


def _conv_bwd(dout, xp, w, same, lin):
    """dout (n, lout, Cout) -> (dx (n, lin, Cin), dw, db)."""
    k = w.shape[0]
    lout = dout.shape[1]
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp)
    for j in range(k):
        dw[j] = np.einsum("nlc,nlo->co", xp[:, j:j + lout, :], dout)
        dxp[:, j:j + lout, :] += dout @ w[j].T
    db = dout.sum(axis=(0, 1))
    if same:
        pl = (k - 1) // 2
        dxp = dxp[:, pl:pl + lin, :]
    return dxp, dw, db


def loss_and_grads(kind, weights, x, y, dropout_mask=None):
    """MSE loss of the mini-batch and its gradient w.r.t. every weight array (Keras get_weights() order)."""
    W = [np.asarray(a, np.float64) for a in weights]
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    n = x.shape[0]
    if kind == "cnn":
        w1, b1, w2, b2, w3, b3, d1, c1, d2, c2, d3, c3 = W
        z1, xp1 = _conv_fwd(x, w1, b1, same=False); a1 = np.maximum(z1, 0)
        z2, xp2 = _conv_fwd(a1, w2, b2, same=True); a2 = np.maximum(z2, 0)
        z3, xp3 = _conv_fwd(a2, w3, b3, same=True); a3 = np.maximum(z3, 0)
        g = a3.max(axis=1)                                           # GlobalMaxPooling1D
        u1 = g @ d1 + c1; h1 = np.maximum(u1, 0)
        u2 = h2 = None
        u2 = h1 @ d2 + c2; h2 = np.maximum(u2, 0)
        keep = np.ones_like(h2) if dropout_mask is None else np.asarray(dropout_mask, np.float64) / (1.0 - DROPOUT)
        hd = h2 * keep
        pred = (hd @ d3 + c3)[:, 0]
        dpred = (2.0 / n) * (pred - y)
        gd3 = hd.T @ dpred[:, None]; gc3 = np.array([dpred.sum()])
        dh2 = (dpred[:, None] @ d3.T) * keep
        du2 = dh2 * (u2 > 0)
        gd2 = h1.T @ du2; gc2 = du2.sum(axis=0)
        du1 = (du2 @ d2.T) * (u1 > 0)
        gd1 = g.T @ du1; gc1 = du1.sum(axis=0)
        dg = du1 @ d1.T                                              # (n, F)
        is_max = (a3 == g[:, None, :])
        da3 = is_max * (dg / is_max.sum(axis=1))[:, None, :]         # ties share the gradient evenly
        dz3 = da3 * (z3 > 0)
        da2, gw3, gb3 = _conv_bwd(dz3, xp3, w3, True, a2.shape[1])
        dz2 = da2 * (z2 > 0)
        da1, gw2, gb2 = _conv_bwd(dz2, xp2, w2, True, a1.shape[1])
        dz1 = da1 * (z1 > 0)
        _, gw1, gb1 = _conv_bwd(dz1, xp1, w1, False, x.shape[1])
        grads = [gw1, gb1, gw2, gb2, gw3, gb3, gd1, gc1, gd2, gc2, gd3, gc3]
    else:
        d1, c1, d2, c2, d3, c3, d4, c4 = W
        f = x.reshape(n, -1)
        u1 = f @ d1 + c1; h1 = np.maximum(u1, 0)
        u2 = h1 @ d2 + c2; h2 = np.maximum(u2, 0)
        u3 = h2 @ d3 + c3; h3 = np.maximum(u3, 0)
        pred = (h3 @ d4 + c4)[:, 0]
        dpred = (2.0 / n) * (pred - y)
        gd4 = h3.T @ dpred[:, None]; gc4 = np.array([dpred.sum()])
        du3 = (dpred[:, None] @ d4.T) * (u3 > 0)
        gd3 = h2.T @ du3; gc3 = du3.sum(axis=0)
        du2 = (du3 @ d3.T) * (u2 > 0)
        gd2 = h1.T @ du2; gc2 = du2.sum(axis=0)
        du1 = (du2 @ d2.T) * (u1 > 0)
        gd1 = f.T @ du1; gc1 = du1.sum(axis=0)
        grads = [gd1, gc1, gd2, gc2, gd3, gc3, gd4, gc4]
    loss = float(np.mean((pred - y) ** 2))
    return loss, grads


def new_state(weights):
    return {"t": 0, "m": [np.zeros(np.shape(w)) for w in weights], "v": [np.zeros(np.shape(w)) for w in weights]}


def adam_step(weights, grads, state):
    """One tf.keras Adam update; returns (new weights (float64), new state)."""
    t = state["t"] + 1
    lr_t = LR * np.sqrt(1.0 - BETA_2 ** t) / (1.0 - BETA_1 ** t)
    new_w, new_m, new_v = [], [], []
    for w, g, m, v in zip(weights, grads, state["m"], state["v"]):
        m = BETA_1 * m + (1.0 - BETA_1) * g
        v = BETA_2 * v + (1.0 - BETA_2) * g * g
        new_w.append(np.asarray(w, np.float64) - lr_t * m / (np.sqrt(v) + EPSILON))
        new_m.append(m)
        new_v.append(v)
    return new_w, {"t": t, "m": new_m, "v": new_v}


def train_step(kind, weights, x, y, state, dropout_mask=None):
    loss, grads = loss_and_grads(kind, weights, x, y, dropout_mask)
    new_w, state = adam_step(weights, grads, state)
    return loss, new_w, state


def glorot_limit(shape):
    """Keras glorot_uniform bound: sqrt(6 / (fan_in + fan_out)); conv kernels count the receptive field."""
    receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return float(np.sqrt(6.0 / (shape[-2] * receptive + shape[-1] * receptive)))
