"""The hand-written training step (flexs_amd/csrc/train_core.h: forward in training mode, MSE, reverse-mode gradients,
Keras-form Adam; `KerasModel.train`, flexs/baselines/models/keras_model.py:49-67 with cnn.py:56 / mlp.py:33 /
global_epistasis_model.py:37) against the float64 NumPy restatement oracle/train_np.py.

CPU tier: the HOST build of the very source the kernels are compiled from (`fx_debug_train_step_host`: threads as loops,
the MFMA as an fmaf chain) -- index arithmetic and gradient algebra of every layer form, for every architecture, several
consecutive steps (optimiser state carried), any slicing of the mini-batch.  GPU tier: `fx_train_fit` on the device
against the same oracle step by step, whole fits against the PyTorch path on the same shuffles, and the public
`train` API end to end."""
import os

import numpy as np
import pytest

from flexs_amd import _native, training
from flexs_amd.baselines import models as bm
from oracle import ref_np, train_np

KIND = {"cnn": 0, "mlp": 1, "ge": 2}
CASES = [  # kind, L, alphabet, F, H, K, rows
    ("mlp", 9, "UGCA", 0, 24, 0, 37), ("ge", 9, "UGCA", 0, 20, 0, 37), ("cnn", 9, "UGCA", 8, 16, 3, 37),
    ("cnn", 8, "TGCA", 32, 100, 5, 24), ("cnn", 12, ref_np.AAS, 5, 7, 4, 19), ("mlp", 14, "UGCA", 0, 100, 0, 40),
    ("ge", 30, ref_np.AAS, 0, 100, 0, 33), ("cnn", 6, "01", 3, 5, 2, 11), ("mlp", 5, ref_np.AAS, 0, 9, 0, 5),
]


def _flat(ws):
    return np.ascontiguousarray(np.concatenate([np.asarray(w, np.float32).ravel() for w in ws]))


def _unflat(flat, shapes):
    out, off = [], 0
    for s in shapes:
        k = int(np.prod(s))
        out.append(flat[off:off + k].reshape(s))
        off += k
    return out


def _shapes(kind, L, A, F, H, K):
    return {"cnn": lambda: ref_np.cnn_shapes(L, A, F, H, K), "mlp": lambda: ref_np.mlp_shapes(L, A, H), "ge": lambda: ref_np.ge_shapes(L, A, H)}[kind]()


def _data(kind, L, alphabet, rows, seed):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(alphabet), (rows, L))
    seqs = ["".join(alphabet[i] for i in row) for row in idx]
    x = ref_np.encode_batch(seqs, alphabet).astype(np.float32)
    y = rng.normal(size=rows).astype(np.float32)
    b = np.frombuffer("".join(seqs).encode(), np.uint8).reshape(rows, L)
    return seqs, b, x, y


def _near_tie_in_a_max_pool(kind, w64, x):
    """True if some GlobalMaxPooling1D of the batch has its two largest activations within 1e-4 relative of each other without
    being EQUAL (equal ones -- repeated windows, common on the binary alphabet -- share the gradient on both sides).  Float32
    forward error (~1e-6 through three conv layers) can then pick the other position than the float64 oracle: a DISCRETE change
    of the routed gradient (~1 / rows of a conv kernel's column), a property of max-pooling, not of the kernel under test."""
    if kind != "cnn":
        return False
    a = x.astype(np.float64)
    for i, same in ((0, False), (2, True), (4, True)):
        a = np.maximum(train_np._conv_fwd(a, w64[i], w64[i + 1], same=same)[0], 0)
    srt = np.sort(a, axis=1)
    if srt.shape[1] < 2:
        return False
    top1, top2 = srt[:, -1, :], srt[:, -2, :]
    gap = np.where(top1 > 0, (top1 - top2) / np.maximum(top1, 1e-300), 1.0)
    return bool(((gap > 0) & (gap < 1e-4)).any())


def check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=3, resync=True):
    """step_fn(w_flat, m_flat, v_flat, t, seq_bytes, y, keep) -> (t', loss); arrays updated in place.

    resync: after every checked step the float32 state is reset to the oracle's (weights, moments), so that each step is
    compared from IDENTICAL inputs.  Chained on its own float32 weights a step can sit on the other side of a ReLU / max-pool
    kink from the float64 oracle (a pre-activation within 1e-8 of zero flips a unit's gradient: seen in the round-3 soak, 2 of
    150 random shapes, reproduced bit for bit by the host build) -- a property of the network, not of the kernel."""
    A = len(alphabet)
    shapes = _shapes(kind, L, A, F, H, K)
    w64 = [a.astype(np.float64) for a in ref_np.synth_weights(shapes, 77)]
    state = train_np.new_state(w64)
    w, m, v, t = _flat(w64), np.zeros(sum(int(np.prod(s)) for s in shapes), np.float32), None, 0
    v = m.copy()
    for step in range(steps):
        _, b, x, y = _data(kind, L, alphabet, rows, 100 + step)
        keep = (np.random.default_rng(step).random((rows, H)) >= train_np.DROPOUT).astype(np.uint8) if kind == "cnn" else None
        mask = None if keep is None else keep.astype(np.float32)
        kinked = _near_tie_in_a_max_pool(kind, w64, x)
        _, grads = train_np.loss_and_grads(kind, w64, x, y, mask)
        want_loss, w64, state = train_np.train_step(kind, w64, x, y, state, mask)
        t, got_loss = step_fn(w, m, v, t, b, y, keep)
        assert t == state["t"] == step + 1
        assert got_loss == pytest.approx(want_loss, rel=3e-5, abs=1e-7)
        lr_t = train_np.LR * np.sqrt(1.0 - train_np.BETA_2 ** t) / (1.0 - train_np.BETA_1 ** t)
        if kinked:                                           # forward (loss) checked; the backward routing is ambiguous in float32
            w[:] = _flat(w64); m[:] = _flat(state["m"]); v[:] = _flat(state["v"])
            continue
        for i, (a, ref, g, vv) in enumerate(zip(_unflat(w, shapes), w64, grads, state["v"])):
            # 2e-6 absolute pins an Adam step (~1e-3) to 0.2 %.  Where a gradient is a near-total cancellation (|g| ~ 1e-7: a
            # first-layer weight whose letter barely occurs), Adam divides its float32 rounding error dg by sqrt(v) + 1e-7
            # ~ 1e-7 and turns it into a visible weight change: those elements get d(update)/dg x dg on top, with dg = 2e-5 of the
            # array's largest gradient -- nothing for a well-conditioned element (sqrt(v) ~ |g|: + 1e-8)
            tol = 2e-6 + 2e-6 * np.abs(ref) + lr_t * (1.0 - train_np.BETA_1) * 2e-5 * np.abs(g).max() / (np.sqrt(vv) + train_np.EPSILON)
            bad = np.abs(a - ref) > tol
            assert not bad.any(), (kind, step, i, float(np.abs(a - ref).max()), int(bad.sum()))
        # moments: float32 sums of up to a few hundred terms of mixed sign against float64 -- the rounding error scales with the
        # terms, not with the (possibly cancelled) result: relative to the array's scale
        for a, ref in zip(_unflat(m, shapes), state["m"]):
            assert np.allclose(a, ref, rtol=3e-4, atol=1e-8 + 3e-5 * np.abs(ref).max()), (kind, step, "m", np.abs(a - ref).max())
        for a, ref in zip(_unflat(v, shapes), state["v"]):
            assert np.allclose(a, ref, rtol=6e-4, atol=1e-12 + 6e-5 * np.abs(ref).max()), (kind, step, "v", np.abs(a - ref).max())
        if resync:
            w[:] = _flat(w64)
            m[:] = _flat(state["m"])
            v[:] = _flat(state["v"])


@pytest.mark.parametrize("kind,L,alphabet,F,H,K,rows", CASES)
@pytest.mark.parametrize("R", [16, 5])
def test_host_build_of_the_training_step_equals_the_keras_restatement(kind, L, alphabet, F, H, K, rows, R):
    lut = _native.make_lut(alphabet)

    def step_fn(w, m, v, t, b, y, keep):
        return _native.debug_train_step_host(KIND[kind], L, len(alphabet), F, H, K, w, m, v, t, b, lut, y, keep, R=R)

    check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows)


PROTEIN_CASES = [  # kind, L, alphabet, F, H, K, rows   (BASELINE configs[3] / configs[4] lengths: AAV 90, GFP 237 / 238, and their neighbours)
    ("cnn", 90, ref_np.AAS, 32, 100, 5, 64), ("cnn", 230, ref_np.AAS, 32, 100, 5, 16), ("cnn", 237, ref_np.AAS, 32, 100, 5, 32),
    ("cnn", 238, ref_np.AAS, 32, 100, 5, 16), ("cnn", 260, ref_np.AAS, 32, 100, 5, 16), ("cnn", 300, ref_np.AAS, 32, 100, 5, 16),
]


@pytest.mark.parametrize("kind,L,alphabet,F,H,K,rows", [PROTEIN_CASES[0], ("cnn", 237, ref_np.AAS, 32, 100, 5, 6)])
def test_host_build_at_protein_lengths_equals_the_keras_restatement(kind, L, alphabet, F, H, K, rows):
    """The host build at the lengths the reference retrains at on the protein landscapes (explorer.py:157-160 with the AAV / GFP
    landscapes' seq_len; cnn.py:23-56 -> conv3 has 19 taps): one row per slice, as the device cuts those fits."""
    lut = _native.make_lut(alphabet)

    def step_fn(w, m, v, t, b, y, keep):
        return _native.debug_train_step_host(KIND[kind], L, len(alphabet), F, H, K, w, m, v, t, b, lut, y, keep, R=1)

    check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=2)


def test_host_build_slicing_does_not_change_the_step():
    """The gradient is a sum over slices in slice order: different R give the same weights to float32 rounding."""
    kind, L, alphabet, F, H, K, rows = "cnn", 8, "TGCA", 8, 12, 3, 50
    lut = _native.make_lut(alphabet)
    shapes = _shapes(kind, L, 4, F, H, K)
    _, b, _, y = _data(kind, L, alphabet, rows, 5)
    outs = []
    for R in (1, 4, 16, 64):
        w = _flat(ref_np.synth_weights(shapes, 3))
        m, v = np.zeros_like(w), np.zeros_like(w)
        _native.debug_train_step_host(0, L, 4, F, H, K, w, m, v, 0, b, lut, y, np.ones((rows, H), np.uint8), R=R)
        outs.append(w)
    for o in outs[1:]:
        assert np.abs(o - outs[0]).max() < 2e-6


# ---------------------------------------------------------------------------------------------------------------
# GPU tier
def _fit_once(eng, kind, L, A, F, H, K, w, m, v, t, b, y, order, epochs, batch, keep=None, seed=0, lut=None):
    (t2, loss), = _native.train_fit(eng, [{"kind": KIND[kind], "L": L, "A": A, "F": F, "H": H, "K": K, "weights": w, "adam_m": m,
                                           "adam_v": v, "step": t, "order": order, "epochs": epochs, "batch": batch,
                                           "keep": keep, "seed": seed}], b, lut, y)
    return t2, loss


@pytest.mark.gpu
@pytest.mark.parametrize("kind,L,alphabet,F,H,K,rows", CASES + [("cnn", 30, ref_np.AAS, 32, 100, 5, 40), ("cnn", 14, "UGCA", 32, 100, 5, 256)])
def test_device_training_step_equals_the_keras_restatement(kind, L, alphabet, F, H, K, rows):
    """fx_train_fit on the MI355X, one step per call (epochs = 1, batch = the whole data set, explicit dropout masks),
    against oracle/train_np.py: loss, every weight within 2e-6 after each Adam step, moments, step count."""
    eng = _native.Engine.get(0)
    lut = _native.make_lut(alphabet)

    def step_fn(w, m, v, t, b, y, keep):
        t2, loss = _fit_once(eng, kind, L, len(alphabet), F, H, K, w, m, v, t, b, y, np.arange(rows, dtype=np.int32), 1, rows,
                             keep=None if keep is None else keep[None], lut=lut)
        return t2, float(loss[0])

    check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=3)


@pytest.mark.gpu
@pytest.mark.parametrize("swizzle", [None, 0, 1, 2, 3])
@pytest.mark.parametrize("kind,L,alphabet,F,H,K,rows", PROTEIN_CASES)
def test_device_training_step_at_protein_lengths_equals_the_keras_restatement(kind, L, alphabet, F, H, K, rows, swizzle):
    """Round 5 (verdict item 1): the device step against oracle/train_np.py -- not only against the plain device step -- at the
    lengths BASELINE configs[3] / configs[4] retrain at: CNN L = 90 / A = 20 (padded rows in LDS, weights from L2), L = 230 / 237 /
    238 (one row per slice, unpadded or rotated rows in LDS, the staged conv kernels), L = 260 (past the five-array LDS layout) and
    L = 300 (workspace in global memory), 16 - 64 rows; under the engine's default form (None) and every explicit `train_swizzle`."""
    eng = _native.Engine.get(0)
    lut = _native.make_lut(alphabet)
    default = eng.get_option("train_swizzle")
    if swizzle is not None:
        eng.set_option("train_swizzle", swizzle)

    def step_fn(w, m, v, t, b, y, keep):
        t2, loss = _fit_once(eng, kind, L, len(alphabet), F, H, K, w, m, v, t, b, y, np.arange(rows, dtype=np.int32), 1, rows,
                             keep=None if keep is None else keep[None], lut=lut)
        return t2, float(loss[0])

    try:
        check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=2)
    finally:
        eng.set_option("train_swizzle", default)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,F,H,K", [("cnn", 8, 16, 3), ("mlp", 0, 24, 0), ("ge", 0, 20, 0)])
def test_device_fit_with_a_partial_last_batch_equals_the_restatement(kind, F, H, K):
    """A whole fit in ONE call: 2 epochs x 3 mini-batches of 128 slots over 300 rows (the last one holds 44 valid rows and 84
    padding slots), given shuffles and masks; the oracle replays the same steps on the rows each mini-batch holds."""
    L, alphabet, n, B, epochs = 9, "UGCA", 300, 128, 2
    A, steps = 4, 3
    eng = _native.Engine.get(0)
    lut = _native.make_lut(alphabet)
    shapes = _shapes(kind, L, A, F, H, K)
    _, b, x, y = _data(kind, L, alphabet, n, 9)
    rng = np.random.default_rng(4)
    order = np.full((epochs, steps * B), -1, np.int32)
    for e in range(epochs):
        order[e, :n] = rng.permutation(n)
    keep = (rng.random((epochs * steps, B, H)) >= train_np.DROPOUT).astype(np.uint8) if kind == "cnn" else None
    w64 = [a.astype(np.float64) for a in ref_np.synth_weights(shapes, 5)]
    state = train_np.new_state(w64)
    w = _flat(w64); m = np.zeros_like(w); v = np.zeros_like(w)
    want_losses = []
    for e in range(epochs):
        for s in range(steps):
            rows = order[e, s * B:(s + 1) * B]
            rows = rows[rows >= 0]
            mask = None if keep is None else keep[e * steps + s, :len(rows)].astype(np.float32)
            loss, w64, state = train_np.train_step(kind, w64, x[rows], y[rows], state, mask)
            want_losses.append(loss)
    t, losses = _fit_once(eng, kind, L, A, F, H, K, w, m, v, 0, b, y, order, epochs, B, keep=keep, lut=lut)
    assert t == epochs * steps == state["t"]
    assert np.allclose(losses, want_losses, rtol=1e-4, atol=1e-7)
    for a, ref in zip(_unflat(w, shapes), w64):
        assert np.abs(a - ref).max() <= 6e-6 + 1e-5 * np.abs(ref).max(), (kind, np.abs(a - ref).max())   # 6 Adam steps
    # characters outside the alphabet are refused before anything runs (ValueError in the reference's encode loop)
    bad = b.copy(); bad[7, 3] = ord("Z")
    with pytest.raises(ValueError):
        _fit_once(eng, kind, L, A, F, H, K, w, m, v, t, bad, y, order, epochs, B, keep=keep, lut=lut)


@pytest.mark.gpu
def test_native_training_through_the_plugin_api_and_its_speed():
    """`Ensemble.train` of the 3-CNN ensemble on 1000 measured sequences (the explorer round of bench.py): one device call,
    every member learns, deterministic for a seed (bit-identical repeat), members trained together == one by one, the
    in-kernel dropout stream keeps ~75 % of the units; wall time printed."""
    import time

    import flexs_amd

    rng = np.random.default_rng(0)
    alphabet, L, n = "TGCA", 8, 1000
    seqs = ["".join(alphabet[i] for i in row) for row in rng.integers(0, 4, (n, L))]
    y = np.array([s.count("G") / L + 0.5 * (s[0] == "T") for s in seqs], np.float32)

    def fresh():
        return flexs_amd.Ensemble([bm.CNN(L, 32, 100, alphabet, seed=m) for m in range(3)])

    ens = fresh()
    before = [float(np.mean((m.get_fitness(seqs) - y) ** 2)) for m in ens.models]
    ens.train(seqs, y, seed=3)
    for m, b0 in zip(ens.models, before):
        assert m.model._opt_state["t"] == 20 * 4
        assert float(np.mean((m.get_fitness(seqs) - y) ** 2)) < 0.3 * b0
    again = fresh(); again.train(seqs, y, seed=3)
    solo = fresh()
    for k, m in enumerate(solo.models):
        m.train(seqs, y, seed=3 + k)
    for a, b_, c in zip(ens.models, again.models, solo.models):
        for wa, wb, wc in zip(a.model.get_weights(), b_.model.get_weights(), c.model.get_weights()):
            assert np.array_equal(wa, wb) and np.array_equal(wa, wc)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ens.train(seqs, y); ts.append(time.perf_counter() - t0)
    print(f"Ensemble.train 3xCNN(32,100) L=8 n=1000 (80 steps per member): {min(ts) * 1e3:.2f} ms")
    # the in-kernel keep stream: fraction kept over one step's (batch x H) draws
    from flexs_amd import training
    assert training._train_mode(__import__("torch").device("cuda")) == "native"


@pytest.mark.gpu
def test_device_training_step_on_a_seeded_sweep_of_random_shapes():
    """24 random draws of (architecture, alphabet, sequence length, filters, hidden width, kernel size, batch rows): two
    consecutive device steps of each against oracle/train_np.py -- whatever tile counts, k-step overhangs, padding taps and
    slice sizes the draw produces (the training kernels are shape-agnostic: every contraction goes through one routine)."""
    eng = _native.Engine.get(0)
    rng = np.random.default_rng(2024)
    for draw in range(24):
        kind = ("cnn", "mlp", "ge")[draw % 3]
        alphabet = ("UGCA", ref_np.AAS, "01", "TGCA")[int(rng.integers(0, 4))]
        A = len(alphabet)
        K = int(rng.integers(2, 8)) if kind == "cnn" else 0
        L = int(rng.integers(max(K, 1), 41))
        F = int(rng.integers(1, 41)) if kind == "cnn" else 0
        H = int(rng.integers(1, 131))
        rows = int(rng.integers(1, 97))
        if kind == "cnn" and A < 2:
            continue
        lut = _native.make_lut(alphabet)

        def step_fn(w, m, v, t, b, y, keep, kind=kind, L=L, A=A, F=F, H=H, K=K, rows=rows, lut=lut):
            t2, loss = _fit_once(eng, kind, L, A, F, H, K, w, m, v, t, b, y, np.arange(rows, dtype=np.int32), 1, rows,
                                 keep=None if keep is None else keep[None], lut=lut)
            return t2, float(loss[0])

        try:
            check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=2)
        except AssertionError as exc:
            raise AssertionError(f"draw {draw}: {kind} L={L} A={A} F={F} H={H} K={K} rows={rows}: {exc}") from exc


@pytest.mark.gpu
@pytest.mark.parametrize("kind,L,alphabet,F,H,K,n,B,M", [("cnn", 8, "TGCA", 32, 100, 5, 1000, 256, 3), ("mlp", 14, "UGCA", 0, 100, 0, 300, 128, 2),
                                                        ("ge", 30, ref_np.AAS, 0, 100, 0, 300, 64, 8), ("cnn", 9, "UGCA", 8, 16, 3, 150, 256, 1),
                                                        ("cnn", 30, ref_np.AAS, 32, 100, 5, 120, 64, 2)])
def test_one_launch_fit_equals_the_launch_per_step_fit(kind, L, alphabet, F, H, K, n, B, M):
    """Round 4: the whole fit as ONE launch (`train_persistent`; measured no faster, so off by default: the (slices x members) workgroups stay for all
    steps, the workgroups of a member meet at two barriers in device memory per step, Adam is applied by the same
    workgroups) against round 3's two launches per step -- the same fxt_forward_backward and fxt_adam, the same slice order
    of the gradient sum: weights, both moments, step count and per-step losses are the SAME BITS, for several members with
    different seeds in one call, a ragged last mini-batch and the in-kernel dropout stream."""
    eng = _native.Engine.get(0)
    A, epochs = len(alphabet), 3
    lut = _native.make_lut(alphabet)
    shapes = _shapes(kind, L, A, F, H, K)
    _, b, _, y = _data(kind, L, alphabet, n, 21)
    steps = (n + B - 1) // B
    rng = np.random.default_rng(8)
    results = []
    for persistent in (1, 0):
        eng.set_option("train_persistent", persistent)
        try:
            jobs = []
            r2 = np.random.default_rng(8)
            for mem in range(M):
                order = np.full((epochs, steps * B), -1, np.int32)
                for e in range(epochs):
                    order[e, :n] = r2.permutation(n)
                w = _flat(ref_np.synth_weights(shapes, 40 + mem))
                jobs.append({"kind": KIND[kind], "L": L, "A": A, "F": F, "H": H, "K": K, "weights": w, "adam_m": np.zeros_like(w),
                             "adam_v": np.zeros_like(w), "step": 0, "order": order, "epochs": epochs, "batch": B, "seed": 1234 + mem})
            res = _native.train_fit(eng, jobs, b, lut, y)
            results.append([(j["weights"].copy(), j["adam_m"].copy(), j["adam_v"].copy(), t, np.asarray(loss).copy()) for j, (t, loss) in zip(jobs, res)])
        finally:
            eng.set_option("train_persistent", 0)
    del rng
    for mem, (a, c) in enumerate(zip(*results)):
        assert a[3] == c[3] == epochs * steps
        for what, x, z in zip(("weights", "adam_m", "adam_v", "losses"), (a[0], a[1], a[2], a[4]), (c[0], c[1], c[2], c[4])):
            assert np.array_equal(x, z), (kind, mem, what, float(np.abs(x - z).max()))
        assert np.isfinite(a[0]).all() and not np.array_equal(a[0], _flat(ref_np.synth_weights(shapes, 40 + mem)))


@pytest.mark.gpu
def test_more_members_than_one_device_call_takes():
    """`Ensemble.train` with 70 like members (round-3 advisor finding: fx_train_fit takes 64 per call and `fit_many` handed it the
    whole group): the group is cut into calls of <= 64, seeds are per member, so every member's weights equal a fit on its own."""
    import flexs_amd

    rng = np.random.default_rng(1)
    alphabet, L, n = "UGCA", 9, 90
    seqs = ["".join(alphabet[i] for i in row) for row in rng.integers(0, 4, (n, L))]
    y = rng.random(n).astype(np.float32)
    ens = flexs_amd.Ensemble([bm.GlobalEpistasisModel(L, 12, alphabet, seed=m, epochs=2, batch_size=32) for m in range(70)])
    ens.train(seqs, y, seed=5)
    for k in (0, 63, 64, 69):
        solo = bm.GlobalEpistasisModel(L, 12, alphabet, seed=k, epochs=2, batch_size=32)
        solo.train(seqs, y, seed=5 + k)
        for wa, wb in zip(ens.models[k].model.get_weights(), solo.model.get_weights()):
            assert np.array_equal(wa, wb), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,L,alphabet,F,H,K,n,B,M", [("cnn", 8, "TGCA", 32, 100, 5, 1000, 256, 3), ("cnn", 14, "UGCA", 32, 100, 5, 300, 256, 1),
                                                        ("mlp", 14, "UGCA", 0, 100, 0, 300, 256, 2), ("ge", 90, ref_np.AAS, 0, 100, 0, 300, 256, 8),
                                                        # round 5: run-time rows per slice (4-letter CNN at other lengths; short and long protein CNNs in the F = 32 form)
                                                        ("cnn", 20, "UGCA", 32, 100, 5, 300, 256, 2), ("cnn", 50, "TGCA", 32, 100, 5, 200, 128, 1),
                                                        ("cnn", 20, ref_np.AAS, 32, 100, 5, 300, 256, 2), ("cnn", 90, ref_np.AAS, 32, 100, 5, 200, 256, 1)])
def test_canonical_shape_instantiations_equal_the_shape_agnostic_step(kind, L, alphabet, F, H, K, n, B, M):
    """Round 4: the canonical surrogates (CNN(32, 100, kernel 5) on 4 letters, MLP(100), GlobalEpistasis(100) on 20 letters) train
    through an instantiation of the SAME source with the dimensions as compile-time constants (`train_canon`, default on: dead
    k-step walks and masks fold away, a quarter of the code) -- the arithmetic and its order are untouched, so weights, moments,
    step count and losses are the SAME BITS as the shape-agnostic instantiation's, with the in-kernel dropout stream, several
    members and a ragged last mini-batch."""
    eng = _native.Engine.get(0)
    A, epochs = len(alphabet), 2
    lut = _native.make_lut(alphabet)
    shapes = _shapes(kind, L, A, F, H, K)
    _, b, _, y = _data(kind, L, alphabet, n, 31)
    steps = (n + B - 1) // B
    results = []
    for canon in (1, 0):
        eng.set_option("train_canon", canon)
        try:
            jobs = []
            r2 = np.random.default_rng(9)
            for mem in range(M):
                order = np.full((epochs, steps * B), -1, np.int32)
                for e in range(epochs):
                    order[e, :n] = r2.permutation(n)
                w = _flat(ref_np.synth_weights(shapes, 50 + mem))
                jobs.append({"kind": KIND[kind], "L": L, "A": A, "F": F, "H": H, "K": K, "weights": w, "adam_m": np.zeros_like(w),
                             "adam_v": np.zeros_like(w), "step": 0, "order": order, "epochs": epochs, "batch": B, "seed": 77 + mem})
            res = _native.train_fit(eng, jobs, b, lut, y)
            results.append([(j["weights"].copy(), j["adam_m"].copy(), j["adam_v"].copy(), t, np.asarray(loss).copy()) for j, (t, loss) in zip(jobs, res)])
        finally:
            eng.set_option("train_canon", 1)
    for mem, (a, c) in enumerate(zip(*results)):
        assert a[3] == c[3] == epochs * steps
        for what, x, z in zip(("weights", "adam_m", "adam_v", "losses"), (a[0], a[1], a[2], a[4]), (c[0], c[1], c[2], c[4])):
            assert np.array_equal(x, z), (kind, mem, what, float(np.abs(x - z).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("L,n,B,M", [(237, 300, 256, 3), (238, 130, 128, 1), (230, 100, 64, 2), (90, 200, 256, 2), (260, 40, 32, 1)])
def test_rotated_rows_and_staged_conv_kernels_equal_the_plain_step(L, n, B, M):
    """Written at the end of round 4, first run on the device in round 5 (`train_swizzle`): GFP-length CNN fits -- unpadded activation rows in LDS, every conv
    operand fetch 16-way bank-conflicted -- with rotated rows (1) and, on top, the gradient array over the last conv output and the
    conv kernels staged through LDS in tap groups (2).  Where a value is stored and which memory a weight is read from do not change
    the arithmetic or its order: weights, moments, step count and losses are the SAME BITS as the plain step's, with the in-kernel
    dropout stream, several members and a ragged last mini-batch."""
    eng = _native.Engine.get(0)
    kind, alphabet, F, H, K = "cnn", ref_np.AAS, 32, 100, 5
    A, epochs = len(alphabet), 2
    lut = _native.make_lut(alphabet)
    shapes = _shapes(kind, L, A, F, H, K)
    _, b, _, y = _data(kind, L, alphabet, n, 31)
    steps = (n + B - 1) // B
    results = []
    default = eng.get_option("train_swizzle")
    for swz in (0, 1, 2, 3):
        eng.set_option("train_swizzle", swz)
        try:
            jobs = []
            r2 = np.random.default_rng(9)
            for mem in range(M):
                order = np.full((epochs, steps * B), -1, np.int32)
                for e in range(epochs):
                    order[e, :n] = r2.permutation(n)
                w = _flat(ref_np.synth_weights(shapes, 50 + mem))
                jobs.append({"kind": KIND[kind], "L": L, "A": A, "F": F, "H": H, "K": K, "weights": w, "adam_m": np.zeros_like(w),
                             "adam_v": np.zeros_like(w), "step": 0, "order": order, "epochs": epochs, "batch": B, "seed": 77 + mem})
            res = _native.train_fit(eng, jobs, b, lut, y)
            results.append([(j["weights"].copy(), j["adam_m"].copy(), j["adam_v"].copy(), t, np.asarray(loss).copy()) for j, (t, loss) in zip(jobs, res)])
        finally:
            eng.set_option("train_swizzle", default)
    for swz in (1, 2, 3):
        for mem, (a, c) in enumerate(zip(results[0], results[swz])):
            assert a[3] == c[3] == epochs * steps
            for what, x, z in zip(("weights", "adam_m", "adam_v", "losses"), (a[0], a[1], a[2], a[4]), (c[0], c[1], c[2], c[4])):
                assert np.isfinite(z).all() and np.array_equal(x, z), (swz, L, mem, what, float(np.abs(x - z).max()))
