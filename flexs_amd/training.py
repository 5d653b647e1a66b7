"""PyTorch replacement of `KerasModel.train` (flexs/baselines/models/keras_model.py:49-67).

SURVEY.md section 8(f)-1: surrogate training is the step right before the hot
path every explorer round, kept in host Python / PyTorch-ROCm.  It reproduces
what `model.compile(loss="MSE", optimizer="adam")` + `model.fit(batch_size=256,
epochs=20)` do (cnn.py:56, mlp.py:33, global_epistasis_model.py:37):

* loss: mean over the mini-batch of (prediction - label)^2;
* optimiser: Keras' Adam, in Keras' own formulation -- step size
  lr * sqrt(1 - b2^t) / (1 - b1^t) and `m / (sqrt(v) + epsilon)` with epsilon =
  1e-7 ("epsilon hat" of Kingma & Ba, not the epsilon of their Algorithm 1, which is
  what torch.optim.Adam implements) -- with lr 1e-3, betas .9 / .999;
* optimiser STATE (first / second moments and the step count t) lives on the
  `Architecture` and carries over from one `train` call to the next, as the
  compiled Keras model's optimiser does across the explorer's rounds
  (flexs/explorer.py:157-160 calls `model.train` once per round on the same
  compiled model);
* a fresh shuffle per epoch, a final partial mini-batch, Dropout(0.25) before
  the CNN's last Dense layer in training mode (cnn.py:51).

Parity with Keras is per step (tests/test_training.py checks one mini-batch step
of every architecture against a NumPy restatement, oracle/train_np.py), not
bitwise over a whole fit: the shuffles and dropout masks come from other RNG
streams.

Round 3: on a GPU the fit runs in hand-written HIP (`fx_train_fit`, csrc/train_core.h +
train.hip): per mini-batch step ONE forward+backward launch over (row slices x ensemble
members) and ONE Adam launch, every contraction on v_mfma_f32_16x16x4_f32, all steps of
all members enqueued back to back from C.  The host only draws the shuffles (one
`fx_train_orders` call per member) and keeps weights / moments / step count
on the `Architecture`.  FLEXS_AMD_TRAIN=graph | eager selects the PyTorch paths below
instead (captured hipGraph step / plain eager step); without a GPU the eager PyTorch step
runs on the CPU (training is not the scored hot path; the trained weights are then
uploaded to the scoring engine).

On the GPU one mini-batch step is ~150 tiny kernels (the networks have 12-40 thousand
parameters and a batch is 256 rows): launched one by one from Python a step costs
2.4 ms for the canonical CNN, 0.19 s per `train` call of an explorer round -- the
GPU idles between launches.  `fit` therefore captures ONE step (gather of the
mini-batch, forward, loss, backward, Adam with the step count on the device) in a
hipGraph (`torch.cuda.CUDAGraph`) and replays it: the mini-batch indices and row
weights are copied into the graph's static buffers before each replay, a final
partial mini-batch is padded with zero-weight rows (the loss is the mean over the
valid rows, as Keras' smaller last batch gives).  Same arithmetic as the eager
step (`_adam_update` is shared; tests hold the two to each other), FLEXS_AMD_TRAIN_GRAPH=0
switches the capture off.

Dropout masks never come from torch's default generator: every `fit` owns a generator on
the training device (seeded from the fit's own shuffle generator) and draws one (batch, H)
Bernoulli mask per step from it -- OUTSIDE the captured graph, into a static buffer the graph
reads.  A captured `F.dropout` would read the Philox offset that `graph.replay()` refills per
generator, and the members of an `Ensemble` replay on separate streams: one member's dropout
kernel could then see the offset written for another (round-2 advisor finding).  With explicit
masks the interleaved and the one-by-one training of CNN members are the same arithmetic.
"""
from __future__ import annotations

import math
import os
import weakref

import numpy as np
import torch
import torch.nn.functional as F

from flexs_amd import _native

LR, BETA_1, BETA_2, EPSILON = 1e-3, 0.9, 0.999, 1e-7      # tf.keras.optimizers.Adam defaults
DROPOUT = 0.25                                            # cnn.py:51


def _codes(sequences, alphabet, L, device):
    """(n, L) uint8 alphabet indices on `device` (ValueError for a character outside the alphabet, as
    `alphabet.index` raises in sequence_utils.py:46)."""
    seq_bytes = _native.sequences_to_bytes(sequences, L=L)
    codes = _native.make_lut(alphabet)[seq_bytes]
    if (codes == 255).any():
        raise ValueError("substring not found")
    return torch.from_numpy(np.ascontiguousarray(codes, np.uint8)).to(device)


def _one_hot(codes, A):
    """(..., L) uint8 -> (..., L, A) float32 one-hot; a compare against arange (no host sync, capturable)."""
    return (codes.unsqueeze(-1) == torch.arange(A, dtype=torch.uint8, device=codes.device)).to(torch.float32)


def _encode(sequences, alphabet, L, device):
    return _one_hot(_codes(sequences, alphabet, L, device), len(alphabet))     # (n, L, A)


def _mask_generator(gen, device):
    """Generator on `device` for one fit's dropout masks, seeded by ONE draw from the fit's shuffle generator (so a seeded
    fit is reproducible end to end, and the captured and the eager path consume `gen` alike)."""
    seed = int(torch.randint(0, 2 ** 62, (1,), generator=gen).item())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def _epoch_orders(gen, n: int, epochs: int) -> np.ndarray:
    """The fit's epoch shuffles, (epochs, n) int32: ONE draw from the fit's generator seeds them all (csrc/train.hip
    fx_train_orders: 60 shuffles of 1000 rows in 0.12 ms; one `torch.randperm` per epoch and member was 0.5 ms of a 3.6 ms
    Ensemble.train).  Every path -- native, captured graph, eager -- takes its orders from here, after the draw that seeds its
    dropout stream, so a seeded fit shuffles identically on all of them."""
    seed = int(torch.randint(0, 2 ** 62, (1,), generator=gen).item())
    return _native.train_orders(seed, n, epochs)


def _draw_mask(out, gen):
    """out (B, H) <- 0/1 keep mask, P(keep) = 1 - DROPOUT, from `gen` on the current stream."""
    return out.bernoulli_(1.0 - DROPOUT, generator=gen)


def _conv(h, w, b, same):
    """Keras Conv1D(strides=1), channels-last, as ONE GEMM on the unfolded input (no MIOpen find step on a fresh box, and
    a handful of kernels per layer instead of a slice + GEMM + add per tap: the step is launch-bound, see `fit`):
    h (n, L, Cin), w (k, Cin, Cout) -> (n, Lout, Cout); 'same' pads (k-1)//2 left, the rest right."""
    k, cin, cout = w.shape
    if same:
        pl = (k - 1) // 2
        h = F.pad(h, (0, 0, pl, k - 1 - pl))
    n, lout = h.shape[0], h.shape[1] - k + 1
    cols = h.unfold(1, k, 1).permute(0, 1, 3, 2).reshape(n * lout, k * cin)      # row (n, t): x[n, t + j, c] at (j, c)
    return torch.addmm(b, cols, w.reshape(k * cin, cout)).view(n, lout, cout)


def forward(kind, params, x, train=False, dropout_mask=None):
    """Differentiable forward with parameters in Keras layout; x (n, L, A) -> (n,).
    dropout_mask (n, H) of 0/1 replaces the random Dropout mask (tests)."""
    if kind == "cnn":
        w1, b1, w2, b2, w3, b3, d1, c1, d2, c2, d3, c3 = params
        h = F.relu(_conv(x, w1, b1, same=False))
        h = F.relu(_conv(h, w2, b2, same=True))
        h = F.relu(_conv(h, w3, b3, same=True))      # MaxPooling1D(1) in between is the identity
        h = h.amax(dim=1)
        h = F.relu(h @ d1 + c1)
        h = F.relu(h @ d2 + c2)
        if dropout_mask is not None:
            h = h * dropout_mask / (1.0 - DROPOUT)   # Keras scales the kept units by 1 / (1 - rate)
        else:
            h = F.dropout(h, DROPOUT, training=train)
        return (h @ d3 + c3)[:, 0]
    d1, c1, d2, c2, d3, c3, d4, c4 = params
    h = x.reshape(x.shape[0], -1)
    h = F.relu(h @ d1 + c1)
    h = F.relu(h @ d2 + c2)
    h = F.relu(h @ d3 + c3)
    return (h @ d4 + c4)[:, 0]


def _adam_update(params, grads, m, v, lr_t, b1=BETA_1, b2=BETA_2, eps=EPSILON):
    """m, v, params <- one Keras-Adam update; lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) as a Python float (eager step) or
    a 0-dim float32 device tensor (captured step: the step count lives on the device)."""
    torch._foreach_mul_(m, b1)
    torch._foreach_add_(m, grads, alpha=1.0 - b1)
    torch._foreach_mul_(v, b2)
    torch._foreach_addcmul_(v, grads, grads, value=1.0 - b2)
    denom = torch._foreach_sqrt(v)
    torch._foreach_add_(denom, eps)
    upd = torch._foreach_div(m, denom)
    torch._foreach_mul_(upd, lr_t)
    torch._foreach_sub_(params, upd)


class KerasAdam:
    """Adam as tf.keras applies it (see the module docstring); state = (t, m, v), exported to / restored
    from NumPy so that it can live on the `Architecture` between `train` calls."""

    def __init__(self, params, state=None, lr=LR, beta_1=BETA_1, beta_2=BETA_2, epsilon=EPSILON):
        self.params = params
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        if state is not None and len(state["m"]) == len(params) and \
                all(tuple(m.shape) == tuple(p.shape) for m, p in zip(state["m"], params)):
            self.t = int(state["t"])
            self.m = [torch.as_tensor(a, dtype=p.dtype, device=p.device).clone() for a, p in zip(state["m"], params)]
            self.v = [torch.as_tensor(a, dtype=p.dtype, device=p.device).clone() for a, p in zip(state["v"], params)]
        else:
            self.t = 0
            self.m = [torch.zeros_like(p) for p in params]
            self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        _adam_update(self.params, [p.grad for p in self.params], self.m, self.v, lr_t, self.b1, self.b2, self.eps)

    def state(self):
        return {"t": self.t, "m": [m.detach().cpu().numpy() for m in self.m],
                "v": [v.detach().cpu().numpy() for v in self.v]}


def train_step(arch, x, y, dropout_mask=None, device=None):
    """ONE mini-batch step on `arch` (weights and optimiser state updated in place): x (n, L, A) one-hot float32,
    y (n,) labels.  Returns the loss before the step.  This is the unit `fit` repeats; tests drive it directly."""
    device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    params = [torch.tensor(w, device=device, requires_grad=True) for w in arch._weights]
    opt = KerasAdam(params, getattr(arch, "_opt_state", None))
    xt = torch.as_tensor(np.asarray(x, np.float32), device=device)
    yt = torch.as_tensor(np.asarray(y, np.float32), device=device)
    mask = None if dropout_mask is None else torch.as_tensor(np.asarray(dropout_mask, np.float32), device=device)
    loss = F.mse_loss(forward(arch.kind, params, xt, train=True, dropout_mask=mask), yt)
    loss.backward()
    opt.step()
    arch.set_weights([p.detach().cpu().numpy() for p in params])
    arch._opt_state = opt.state()
    return float(loss.detach())


class _GraphTrainer:
    """One captured mini-batch step of one `Architecture` (see the module docstring).  Everything the graph touches has a
    fixed address: parameters, Adam moments, step count, the data set (capacity rows), the batch's row indices and
    row weights."""

    def __init__(self, arch, device, batch_size, capacity):
        self.kind, self.B, self.cap, self.device = arch.kind, int(batch_size), int(capacity), device
        self.device_index = torch.cuda.current_device()
        self.shapes = [tuple(s) for s in arch.shapes()]
        self.params = [torch.zeros(s, dtype=torch.float32, device=device, requires_grad=True) for s in self.shapes]
        self.m = [torch.zeros(s, dtype=torch.float32, device=device) for s in self.shapes]
        self.v = [torch.zeros(s, dtype=torch.float32, device=device) for s in self.shapes]
        self.t = torch.zeros((), dtype=torch.float64, device=device)
        # the data set as alphabet indices (1 byte per position; the float32 one-hot rows of a 16k-row protein data set
        # would be 310 MB per member, held for the Architecture's lifetime): one-hot is rebuilt per mini-batch in the graph
        self.A = arch.A
        self.x_all = torch.zeros((self.cap, arch.L), dtype=torch.uint8, device=device)
        self.y_all = torch.zeros((self.cap,), dtype=torch.float32, device=device)
        self.idx = torch.zeros((self.B,), dtype=torch.int64, device=device)
        self.wts = torch.ones((self.B,), dtype=torch.float32, device=device)
        self.sq_err = torch.zeros((), dtype=torch.float32, device=device)      # sum of squared errors since last reset
        # CNN: this step's dropout keep mask, drawn by the caller before each replay (never inside the graph)
        self.mask = torch.ones((self.B, arch.H), dtype=torch.float32, device=device) if arch.kind == "cnn" else None
        # warm-up on a side stream (allocator, rocBLAS handles), then capture; neither may leave a trace in the state
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()

    def matches(self, arch, batch_size, n):
        return self.kind == arch.kind and self.B == int(batch_size) and n <= self.cap and \
            self.device_index == torch.cuda.current_device() and self.shapes == [tuple(s) for s in arch.shapes()] and tuple(self.x_all.shape[1:]) == (arch.L,) and self.A == arch.A

    def _step(self):
        for p in self.params:
            p.grad = None
        xb = _one_hot(self.x_all.index_select(0, self.idx), self.A)
        yb = self.y_all.index_select(0, self.idx)
        se = (forward(self.kind, self.params, xb, train=True, dropout_mask=self.mask) - yb) ** 2 * self.wts
        loss = se.sum() / self.wts.sum()                 # mean over the valid rows of the mini-batch
        loss.backward()
        with torch.no_grad():
            self.sq_err += se.sum()
            self.t += 1
            lr_t = (LR * torch.sqrt(1.0 - BETA_2 ** self.t) / (1.0 - BETA_1 ** self.t)).to(torch.float32)
            _adam_update(self.params, [p.grad for p in self.params], self.m, self.v, lr_t)

    @torch.no_grad()
    def load(self, weights, state):
        for p, w in zip(self.params, weights):
            p.copy_(torch.from_numpy(np.ascontiguousarray(w, np.float32)))
        ok = state is not None and len(state["m"]) == len(self.shapes) and \
            all(tuple(a.shape) == s for a, s in zip(state["m"], self.shapes))
        self.t.fill_(float(state["t"]) if ok else 0.0)
        for dst, key in ((self.m, "m"), (self.v, "v")):
            for d, i in zip(dst, range(len(self.shapes))):
                if ok:
                    d.copy_(torch.from_numpy(np.ascontiguousarray(state[key][i], np.float32)))
                else:
                    d.zero_()
        self.sq_err.zero_()

    def state(self):
        return {"t": int(round(float(self.t.item()))), "m": [a.detach().cpu().numpy() for a in self.m],
                "v": [a.detach().cpu().numpy() for a in self.v]}


class _CaptureFailed(RuntimeError):
    pass


_warned = set()


def _warn_once(msg):
    if msg not in _warned:
        _warned.add(msg)
        import warnings

        warnings.warn(msg, RuntimeWarning, stacklevel=3)


_TRAINERS = weakref.WeakKeyDictionary()          # Architecture -> _GraphTrainer (device state stays out of pickles / copies)


def _use_graph(device):
    return _train_mode(device) == "graph"


# ---------------------------------------------------------------------------------------------------------------
# native path: fx_train_fit (csrc/train.hip)
_KIND = {"cnn": _native.FX_CNN, "mlp": _native.FX_MLP, "ge": _native.FX_GE}


def _train_mode(device):
    """'native' (hand-written HIP, default on a GPU) | 'graph' (captured PyTorch step) | 'eager' (PyTorch step by step)."""
    if device.type != "cuda":
        return "eager"
    mode = os.environ.get("FLEXS_AMD_TRAIN")
    if mode is None:                                      # (round-2 switch, still honoured: 0 = eager PyTorch)
        mode = "eager" if os.environ.get("FLEXS_AMD_TRAIN_GRAPH", "1") == "0" else "native"
    return mode if mode in ("native", "graph", "eager") else "native"


def _flat_state(arch):
    """(weights, m, v, t) as flat float32 arrays in Keras get_weights() order (fresh optimiser: zeros, t = 0)."""
    w = np.ascontiguousarray(np.concatenate([a.ravel() for a in arch._weights]).astype(np.float32))
    st = getattr(arch, "_opt_state", None)
    shapes = [tuple(sh) for sh in arch.shapes()]
    ok = st is not None and len(st["m"]) == len(shapes) and all(tuple(a.shape) == sh for a, sh in zip(st["m"], shapes))
    if ok:
        m = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float32).ravel() for a in st["m"]]))
        v = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float32).ravel() for a in st["v"]]))
        return w, m, v, int(st["t"])
    return w, np.zeros_like(w), np.zeros_like(w), 0


def _unflat(flat, shapes):
    out, off = [], 0
    for sh in shapes:
        k = math.prod(sh)                                 # (np.prod costs 7 us per call: 0.75 ms of a 3-member fit)
        out.append(flat[off:off + k].reshape(sh))          # (views: the flat array is this fit's own and lives on in them)
        off += k
    return out


_NATIVE_MAX_MEMBERS = 64          # members per fx_train_fit call (csrc/train.hip)


def _fit_native(archs, sequences, labels, alphabet, batch_sizes, epochs, verbose, gens):
    """One `fx_train_fit` call for members that share the alphabet and the sequence length: every member's shuffles come
    from its own generator (`_epoch_orders`, after the one draw that seeds its dropout stream -- the same
    consumption as the PyTorch paths, so a seeded fit shuffles identically on every path)."""
    n, L = len(sequences), archs[0].L
    seq_bytes = _native.sequences_to_bytes(sequences, L=L)
    lut = _native.make_lut(alphabet)
    if (lut[seq_bytes] == 255).any():
        raise ValueError("substring not found")
    y = np.asarray(labels, dtype=np.float32)
    jobs, flats = [], []
    for arch, B, ep, gen in zip(archs, batch_sizes, epochs, gens):
        B = int(B)
        steps = (n + B - 1) // B
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=gen).item())
        order = np.full((ep, steps * B), -1, np.int32)
        order[:, :n] = _epoch_orders(gen, n, ep)
        w, m, v, t = _flat_state(arch)
        flats.append((w, m, v))
        jobs.append({"kind": _KIND[arch.kind], "L": L, "A": arch.A, "F": arch.F, "H": arch.H, "K": arch.K, "weights": w,
                     "adam_m": m, "adam_v": v, "step": t, "order": order, "epochs": ep, "batch": B, "seed": seed})
    eng = _native.Engine.get(torch.cuda.current_device())
    res = _native.train_fit(eng, jobs, seq_bytes, lut, y)
    for arch, (w, m, v), (t, loss), job in zip(archs, flats, res, jobs):
        shapes = [tuple(sh) for sh in arch.shapes()]
        arch.set_weights(_unflat(w, shapes))
        arch._opt_state = {"t": t, "m": _unflat(m, shapes), "v": _unflat(v, shapes)}
        if verbose:
            B, ep = job["batch"], job["epochs"]
            steps = (n + B - 1) // B
            valid = np.minimum(B, n - np.arange(steps) * B)
            for e_ in range(ep):
                print(f"Epoch {e_ + 1}/{ep} - loss: {float((loss[e_ * steps:(e_ + 1) * steps] * valid).sum() / n):.6f}")


def fit(arch, sequences, labels, alphabet, batch_size=256, epochs=20, verbose=False, seed=None):
    if arch.loss not in ("MSE", "mse", "mean_squared_error"):
        raise ValueError(f"unsupported loss {arch.loss!r} (the reference only ever uses 'MSE')")
    n = len(sequences)
    if n == 0:
        return
    device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    gen = torch.Generator(device="cpu")
    if seed is not None:
        gen.manual_seed(seed)
    if _train_mode(device) == "native":
        return _fit_native([arch], sequences, labels, alphabet, [batch_size], [epochs], verbose, [gen])
    codes = _codes(sequences, alphabet, arch.L, device)
    y = torch.as_tensor(np.asarray(labels, dtype=np.float32), device=device)
    if _use_graph(device):
        try:
            return _fit_graphed(arch, codes, y, n, int(batch_size), epochs, verbose, gen, device)
        except _CaptureFailed as ex:                     # (a driver / PyTorch build that cannot capture this step)
            _warn_once(f"flexs_amd.training: hipGraph capture of the training step failed ({ex.__cause__!r}); training eagerly")
    x = _one_hot(codes, arch.A)
    params = [torch.tensor(w, device=device, requires_grad=True) for w in arch._weights]
    opt = KerasAdam(params, getattr(arch, "_opt_state", None))       # moments and step count of the previous rounds
    mask_gen = _mask_generator(gen, device)
    orders = _epoch_orders(gen, n, epochs)
    mask_buf = torch.ones((int(batch_size), arch.H), dtype=torch.float32, device=device) if arch.kind == "cnn" else None
    for epoch in range(epochs):
        perm = torch.from_numpy(orders[epoch].astype(np.int64)).to(device)
        total = 0.0
        for i in range(0, n, batch_size):
            idx = perm[i:i + batch_size]
            for p in params:
                p.grad = None
            # a whole (batch, H) mask per step, as the captured step draws it; a partial last batch uses its first rows
            mask = _draw_mask(mask_buf, mask_gen)[:idx.shape[0]] if mask_buf is not None else None
            loss = F.mse_loss(forward(arch.kind, params, x[idx], train=True, dropout_mask=mask), y[idx])
            loss.backward()
            opt.step()
            total += float(loss.detach()) * idx.shape[0]
        if verbose:
            print(f"Epoch {epoch + 1}/{epochs} - loss: {total / n:.6f}")
    arch.set_weights([p.detach().cpu().numpy() for p in params])
    arch._opt_state = opt.state()


class _GraphedFit:
    """One `fit` on the captured step, cut into the pieces `fit_many` interleaves: begin (weights, optimiser state and
    data into the static buffers), start_epoch (a fresh shuffle), step (one replay), end (weights back)."""

    def __init__(self, arch, x, y, n, B, epochs, verbose, gen, device, stream=None):
        self.arch, self.n, self.B, self.epochs, self.verbose, self.gen, self.device = arch, n, B, epochs, verbose, gen, device
        self.stream = stream
        tr = _TRAINERS.get(arch)
        if tr is None or not tr.matches(arch, B, n):
            cap = 1024
            while cap < n:
                cap *= 2
            try:
                tr = _GraphTrainer(arch, device, B, cap)
            except Exception as ex:                      # noqa: BLE001 -- whatever the capture raised; the caller trains eagerly
                raise _CaptureFailed(str(ex)) from ex
            _TRAINERS[arch] = tr
        self.tr = tr
        self.mask_gen = _mask_generator(gen, device)
        self.orders = _epoch_orders(gen, n, epochs)
        self.steps = (n + B - 1) // B
        with torch.no_grad():
            tr.load(arch._weights, getattr(arch, "_opt_state", None))
            tr.x_all[:n].copy_(x)
            tr.y_all[:n].copy_(y)
            self.wts = torch.ones((self.steps * B,), dtype=torch.float32, device=device)
            self.wts[n:] = 0.0                           # padding rows of the last mini-batch (they gather row 0)
        self.perm = None
        self.epoch = 0

    def start_epoch(self):
        perm = torch.zeros((self.steps * self.B,), dtype=torch.int64)
        perm[:self.n] = torch.from_numpy(self.orders[self.epoch].astype(np.int64))
        self.perm = perm.to(self.device, non_blocking=True)

    @torch.no_grad()
    def step(self, i):
        lo = i * self.B
        self.tr.idx.copy_(self.perm[lo:lo + self.B])
        self.tr.wts.copy_(self.wts[lo:lo + self.B])
        if self.tr.mask is not None:
            _draw_mask(self.tr.mask, self.mask_gen)          # this job's generator, this job's stream
        self.tr.graph.replay()

    def end_epoch(self):
        self.epoch += 1
        if self.verbose:
            print(f"Epoch {self.epoch}/{self.epochs} - loss: {float(self.tr.sq_err.item()) / self.n:.6f}")
            self.tr.sq_err.zero_()

    def end(self):
        self.arch.set_weights([p.detach().cpu().numpy() for p in self.tr.params])
        self.arch._opt_state = self.tr.state()


def _fit_graphed(arch, x, y, n, B, epochs, verbose, gen, device):
    job = _GraphedFit(arch, x, y, n, B, epochs, verbose, gen, device)
    for _ in range(epochs):
        job.start_epoch()
        for i in range(job.steps):
            job.step(i)
        job.end_epoch()
    job.end()


def fit_many(archs, sequences, labels, alphabets, batch_sizes, epochs, verbose=False, seeds=None):
    """`for model in models: model.train(sequences, labels)` (flexs/ensemble.py:42-52) for members that are all device
    surrogates: the members' captured steps are replayed on one HIP stream per member, step by step in turn, so that
    the GPU works on all members at once -- a captured step is a serial chain of ~100 tiny kernels that leaves the
    machine empty, and three chains side by side take about as long as one.  Every member keeps its own shuffles,
    dropout masks and optimiser state exactly as in the one-by-one loop; only the wall time changes."""
    n = len(sequences)
    device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    if n == 0:
        return
    def one_by_one():
        for k, (arch, alphabet, bs, ep) in enumerate(zip(archs, alphabets, batch_sizes, epochs)):
            fit(arch, sequences, labels, alphabet, batch_size=bs, epochs=ep, verbose=verbose, seed=seeds[k] if seeds else None)

    # (the same Architecture listed twice is trained twice in a row by the reference's loop: not interleavable)
    if len(archs) < 2 or len({id(a) for a in archs}) < len(archs) or _train_mode(device) == "eager":
        return one_by_one()
    if _train_mode(device) == "native":
        for arch in archs:
            if arch.loss not in ("MSE", "mse", "mean_squared_error"):
                raise ValueError(f"unsupported loss {arch.loss!r} (the reference only ever uses 'MSE')")
        groups = {}
        for k, (arch, alphabet) in enumerate(zip(archs, alphabets)):
            groups.setdefault((alphabet, arch.L), []).append(k)
        for (alphabet, _), ks in groups.items():          # members that see the same bytes train in ONE device call
            gens = []
            for k in ks:
                g = torch.Generator(device="cpu")
                if seeds:
                    g.manual_seed(seeds[k])
                gens.append(g)
            # fx_train_fit takes at most _NATIVE_MAX_MEMBERS members per call (train.hip); seeds are per member, so cutting
            # a large group into several calls changes nothing but the number of launches
            for c0 in range(0, len(ks), _NATIVE_MAX_MEMBERS):
                kc = ks[c0:c0 + _NATIVE_MAX_MEMBERS]
                _fit_native([archs[k] for k in kc], sequences, labels, alphabet, [batch_sizes[k] for k in kc],
                            [epochs[k] for k in kc], verbose, gens[c0:c0 + _NATIVE_MAX_MEMBERS])
        return
    for arch in archs:
        if arch.loss not in ("MSE", "mse", "mean_squared_error"):
            raise ValueError(f"unsupported loss {arch.loss!r} (the reference only ever uses 'MSE')")
    y = torch.as_tensor(np.asarray(labels, dtype=np.float32), device=device)
    cur = torch.cuda.current_stream(device)
    jobs, encoded = [], {}
    for k, (arch, alphabet, bs, ep) in enumerate(zip(archs, alphabets, batch_sizes, epochs)):
        key = (alphabet, arch.L)
        if key not in encoded:
            encoded[key] = _codes(sequences, alphabet, arch.L, device)
        gen = torch.Generator(device="cpu")
        if seeds:
            gen.manual_seed(seeds[k])
        try:
            jobs.append(_GraphedFit(arch, encoded[key], y, n, int(bs), ep, verbose, gen, device, stream=torch.cuda.Stream(device=device)))
        except _CaptureFailed:
            return one_by_one()                          # (nothing has been trained yet: `fit` warns and trains eagerly)
    for job in jobs:
        job.stream.wait_stream(cur)                      # the static buffers were filled on the current stream
    for epoch in range(max(j.epochs for j in jobs)):
        live = [j for j in jobs if epoch < j.epochs]
        for job in live:
            with torch.cuda.stream(job.stream):
                job.start_epoch()
        for i in range(max(j.steps for j in live)):
            for job in live:
                if i < job.steps:
                    with torch.cuda.stream(job.stream):
                        job.step(i)
        for job in live:
            with torch.cuda.stream(job.stream):
                job.end_epoch()
    for job in jobs:
        cur.wait_stream(job.stream)
    for job in jobs:
        job.end()
