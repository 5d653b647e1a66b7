"""PyTorch replacement of `KerasModel.train` (flexs/baselines/models/keras_model.py:49-67).

SURVEY.md section 8(f)-1: surrogate training is the step right before the hot
path every explorer round, kept in host Python / PyTorch-ROCm.  It reproduces
what `model.compile(loss="MSE", optimizer="adam")` + `model.fit(batch_size=256,
epochs=20)` do (cnn.py:56, mlp.py:33, global_epistasis_model.py:37): Adam with
Keras defaults (lr 1e-3, betas .9/.999, eps 1e-7), mean-squared error, a fresh
shuffle per epoch, Dropout(0.25) before the CNN's last Dense (cnn.py:51).
Parity with Keras is statistical, not bitwise (different RNG streams).

Runs on `cuda` when a GPU is visible, else on the CPU (training is not the
scored hot path; the trained weights are then uploaded to the scoring engine).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from flexs_amd import _native


def _encode(sequences, alphabet, L, device):
    seq_bytes = _native.sequences_to_bytes(sequences, L=L)
    lut = torch.from_numpy(_native.make_lut(alphabet).astype(np.int64))
    codes = lut[torch.from_numpy(seq_bytes.astype(np.int64))]
    if (codes == 255).any():
        raise ValueError("substring not found")
    return F.one_hot(codes, len(alphabet)).to(torch.float32).to(device)       # (n, L, A)


def _conv(h, w, b, same):
    """Keras Conv1D(strides=1) as K small GEMMs (channels-last, no MIOpen find step on a fresh box):
    h (n, L, Cin), w (k, Cin, Cout) -> (n, Lout, Cout); 'same' pads (k-1)//2 left, the rest right."""
    k = w.shape[0]
    if same:
        pl = (k - 1) // 2
        h = F.pad(h, (0, 0, pl, k - 1 - pl))
    lout = h.shape[1] - k + 1
    out = b.expand(h.shape[0], lout, w.shape[2])
    for j in range(k):
        out = out + h[:, j:j + lout, :] @ w[j]
    return out


def forward(kind, params, x, train=False):
    """Differentiable forward with parameters in Keras layout; x (n, L, A) -> (n,)."""
    if kind == "cnn":
        w1, b1, w2, b2, w3, b3, d1, c1, d2, c2, d3, c3 = params
        h = F.relu(_conv(x, w1, b1, same=False))
        h = F.relu(_conv(h, w2, b2, same=True))
        h = F.relu(_conv(h, w3, b3, same=True))      # MaxPooling1D(1) in between is the identity
        h = h.amax(dim=1)
        h = F.relu(h @ d1 + c1)
        h = F.relu(h @ d2 + c2)
        h = F.dropout(h, 0.25, training=train)
        return (h @ d3 + c3)[:, 0]
    d1, c1, d2, c2, d3, c3, d4, c4 = params
    h = x.reshape(x.shape[0], -1)
    h = F.relu(h @ d1 + c1)
    h = F.relu(h @ d2 + c2)
    h = F.relu(h @ d3 + c3)
    return (h @ d4 + c4)[:, 0]


def fit(arch, sequences, labels, alphabet, batch_size=256, epochs=20, verbose=False, seed=None):
    if arch.loss not in ("MSE", "mse", "mean_squared_error"):
        raise ValueError(f"unsupported loss {arch.loss!r} (the reference only ever uses 'MSE')")
    n = len(sequences)
    if n == 0:
        return
    device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    x = _encode(sequences, alphabet, arch.L, device)
    y = torch.as_tensor(np.asarray(labels, dtype=np.float32), device=device)
    params = [torch.tensor(w, device=device, requires_grad=True) for w in arch._weights]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-7)
    gen = torch.Generator(device="cpu")
    if seed is not None:
        gen.manual_seed(seed)
    for epoch in range(epochs):
        perm = torch.randperm(n, generator=gen).to(device)
        total = 0.0
        for i in range(0, n, batch_size):
            idx = perm[i:i + batch_size]
            opt.zero_grad(set_to_none=True)
            loss = F.mse_loss(forward(arch.kind, params, x[idx], train=True), y[idx])
            loss.backward()
            opt.step()
            total += float(loss.detach()) * idx.shape[0]
        if verbose:
            print(f"Epoch {epoch + 1}/{epochs} - loss: {total / n:.6f}")
    arch.set_weights([p.detach().cpu().numpy() for p in params])
