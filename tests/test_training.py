"""`KerasModel.train` replacement (flexs_amd/training.py) against the NumPy restatement of one Keras training step
(oracle/train_np.py: MSE, hand-written gradients, tf.keras Adam in Keras' epsilon form, state persisting across calls).
Reference: flexs/baselines/models/keras_model.py:49-67 with the compile() calls at cnn.py:56, mlp.py:33,
global_epistasis_model.py:37; one `train` call per explorer round on the same compiled model (explorer.py:157-160)."""
import copy
import pickle

import numpy as np
import pytest

from flexs_amd import training
from flexs_amd.baselines import models as bm
from oracle import ref_np, train_np


def _batch(kind, L, alphabet, n, seed):
    rng = np.random.default_rng(seed)
    seqs = ["".join(alphabet[i] for i in row) for row in rng.integers(0, len(alphabet), (n, L))]
    x = ref_np.encode_batch(seqs, alphabet).astype(np.float32)
    y = rng.normal(size=n).astype(np.float32)
    return seqs, x, y


def _model(kind, L, alphabet, seed):
    if kind == "cnn":
        m = bm.CNN(L, 8, 16, alphabet, kernel_size=3, seed=seed)
    elif kind == "mlp":
        m = bm.MLP(L, 24, alphabet, seed=seed)
    else:
        m = bm.GlobalEpistasisModel(L, 20, alphabet, seed=seed)
    # non-zero biases so that every bias gradient is exercised
    m.model.set_weights(ref_np.synth_weights(m.model.shapes(), 40 + seed))
    return m


def _check_steps(kind, device=None):
    L, alphabet, n = 9, "UGCA", 37
    model = _model(kind, L, alphabet, 3)
    arch = model.model
    w = [a.astype(np.float64) for a in arch.get_weights()]
    state = train_np.new_state(w)
    for step in range(4):                                   # optimiser state must carry from step to step
        _, x, y = _batch(kind, L, alphabet, n, 100 + step)
        mask = (np.random.default_rng(step).random((n, arch.H)) >= train_np.DROPOUT).astype(np.float32) if kind == "cnn" else None
        want_loss, w, state = train_np.train_step(kind, w, x, y, state, mask)
        got_loss = training.train_step(arch, x, y, dropout_mask=mask, device=device)
        assert got_loss == pytest.approx(want_loss, rel=2e-5, abs=1e-7)
        assert arch._opt_state["t"] == state["t"] == step + 1
        for i, (a, b) in enumerate(zip(arch.get_weights(), w)):
            # float32 arithmetic against float64: an Adam step moves every weight by ~lr = 1e-3, so agreement to 2e-6
            # absolute pins the step to 0.2 % -- a torch-style epsilon placement or a reset step count fails by >> that
            assert np.abs(a - b).max() <= 2e-6 + 2e-6 * np.abs(b).max(), (kind, step, i, np.abs(a - b).max())
        for a, b in zip(arch._opt_state["m"], state["m"]):
            assert np.allclose(a, b, rtol=2e-4, atol=1e-8)
        for a, b in zip(arch._opt_state["v"], state["v"]):
            assert np.allclose(a, b, rtol=4e-4, atol=1e-12)
    return arch


@pytest.mark.parametrize("kind", ["mlp", "ge", "cnn"])
def test_one_step_equals_keras_restatement(kind):
    import torch

    _check_steps(kind, device=torch.device("cpu"))


def test_first_adam_step_has_keras_size():
    """t = 1: m / (sqrt(v) + eps) = (1-b1) g / (sqrt(1-b2) |g| + eps) and lr_t = lr sqrt(1-b2) / (1-b1), i.e. every weight
    with a non-negligible gradient moves by almost exactly lr -- and by visibly less than lr where |g| ~ eps / sqrt(1-b2),
    which is where Keras' epsilon placement differs from Algorithm 1 of the Adam paper (torch.optim.Adam)."""
    g = np.array([1.0, -3.0, 1e-3, 1e-6, 1e-8, 0.0])
    new_w, st = train_np.adam_step([np.zeros(6)], [g], train_np.new_state([np.zeros(6)]))
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = -lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7)
    assert np.allclose(new_w[0], want, rtol=1e-12) and st["t"] == 1
    assert np.allclose(new_w[0][:3], -1e-3 * np.sign(g[:3]), rtol=4e-3)
    torch_form = -1e-3 * g / (np.abs(g) + 1e-7)               # Algorithm 1 epsilon: (m / (1-b1)) / (sqrt(v / (1-b2)) + eps)
    assert abs(new_w[0][3] - torch_form[3]) > 0.5e-3          # the two conventions are far apart for tiny gradients


def test_optimizer_state_persists_across_train_calls_and_copies():
    L, alphabet = 8, "TGCA"
    seqs, _, y = _batch("mlp", L, alphabet, 300, 5)
    model = bm.MLP(L, 16, alphabet, seed=1, epochs=2, batch_size=128)
    assert getattr(model.model, "_opt_state", None) is None
    model.train(seqs, y)
    st1 = model.model._opt_state
    assert st1["t"] == 2 * 3                                  # 2 epochs x ceil(300 / 128) mini-batches
    model.train(seqs, y)                                      # next explorer round: same optimiser
    assert model.model._opt_state["t"] == 12
    assert not np.allclose(st1["m"][0], model.model._opt_state["m"][0])
    for clone in (copy.deepcopy(model), pickle.loads(pickle.dumps(model))):
        assert clone.model._opt_state["t"] == 12
        assert all(np.array_equal(a, b) for a, b in zip(clone.model._opt_state["v"], model.model._opt_state["v"]))
    # a set_weights from outside (e.g. weights shipped from another rank) keeps the optimiser, as Keras does
    model.model.set_weights(model.model.get_weights())
    assert model.model._opt_state["t"] == 12
    # an empty training set is a no-op (keras_model.py trains on whatever the explorer has measured)
    model.train([], np.zeros(0))
    assert model.model._opt_state["t"] == 12


def test_glorot_bounds_and_zero_biases():
    """Keras defaults the reference relies on (no initializer arguments in cnn.py / mlp.py): glorot_uniform kernels,
    zero biases."""
    for model in (bm.CNN(14, 32, 100, "UGCA", seed=0), bm.MLP(14, 100, "UGCA", seed=0), bm.GlobalEpistasisModel(90, 100, ref_np.AAS, seed=0)):
        for w, shp in zip(model.model.get_weights(), model.model.shapes()):
            assert w.shape == tuple(shp) and w.dtype == np.float32
            if len(shp) == 1:
                assert not w.any()
            else:
                lim = train_np.glorot_limit(shp)
                assert np.abs(w).max() <= lim and (np.abs(w).max() > 0.8 * lim or w.size < 64)


def test_gradients_of_the_restatement_against_finite_differences():
    """The hand-written backward pass of oracle/train_np.py against central differences (float64)."""
    for kind in ("mlp", "cnn"):
        L, alphabet, n = 7, "TGCA", 5
        model = _model(kind, L, alphabet, 9)
        w = [a.astype(np.float64) for a in model.model.get_weights()]
        _, x, y = _batch(kind, L, alphabet, n, 77)
        mask = (np.random.default_rng(3).random((n, model.model.H)) >= 0.25).astype(np.float64) if kind == "cnn" else None
        _, grads = train_np.loss_and_grads(kind, w, x, y, mask)
        rng = np.random.default_rng(0)
        for i in range(len(w)):
            for _ in range(3):
                idx = tuple(rng.integers(0, s) for s in w[i].shape)
                h = 1e-6
                wp = [a.copy() for a in w]; wp[i][idx] += h
                wm = [a.copy() for a in w]; wm[i][idx] -= h
                num = (train_np.loss_and_grads(kind, wp, x, y, mask)[0] - train_np.loss_and_grads(kind, wm, x, y, mask)[0]) / (2 * h)
                assert num == pytest.approx(grads[i][idx], rel=2e-4, abs=2e-7), (kind, i, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mlp", "ge", "cnn"])
def test_one_step_on_the_gpu_equals_keras_restatement(kind):
    import torch

    arch = _check_steps(kind, device=torch.device("cuda"))
    # the trained weights reach the scoring engine: the next get_fitness uses them
    model = _model(kind, 9, "UGCA", 3)
    model.model.set_weights(arch.get_weights())
    seqs, _, _ = _batch(kind, 9, "UGCA", 50, 1)
    got = model.get_fitness(seqs)
    want = ref_np.keras_fitness(seqs, "UGCA", kind, arch.get_weights(), exact=True)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind,fast", [("mlp", "graph"), ("ge", "graph"), ("cnn", "graph"), ("mlp", "native"), ("ge", "native")])
def test_fast_fit_equals_the_eager_fit(kind, fast, monkeypatch):
    """`fit` on the GPU runs the hand-written HIP step (FLEXS_AMD_TRAIN=native, the default: csrc/train_core.h) or replays
    ONE captured PyTorch step (graph) instead of launching ~150 kernels per step from Python (eager).
    Same shuffles (seeded), same arithmetic: weights and optimiser state after several epochs -- with a partial last
    mini-batch, over several `train` calls (optimiser state carried) and with a growing data set -- equal the eager path's
    to float32 rounding.  The CNN is compared on the captured path only, with Dropout switched off (the three paths draw
    their masks from different streams; the native CNN step is held to the oracle with explicit masks in
    tests/test_train_native.py)."""
    import time

    import torch

    monkeypatch.setattr(training, "DROPOUT", 0.0)
    L, alphabet = 9, "UGCA"
    results = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("FLEXS_AMD_TRAIN", fast if graph == "1" else "eager")
        model = _model(kind, L, alphabet, 3)
        arch = model.model
        for rnd, n in enumerate((300, 700, 1100)):            # 256 + 44 rows; ...; beyond the first capture's 1024 rows
            seqs, _, y = _batch(kind, L, alphabet, n, 50 + rnd)
            training.fit(arch, seqs, y, alphabet, batch_size=256, epochs=3, seed=rnd)
        assert arch._opt_state["t"] == 3 * (2 + 3 + 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        training.fit(arch, seqs, y, alphabet, batch_size=256, epochs=3, seed=9)
        torch.cuda.synchronize()
        results[graph] = (arch.get_weights(), arch._opt_state, time.perf_counter() - t0)
    (wg, sg, tg), (we, se, te) = results["1"], results["0"]
    assert sg["t"] == se["t"]
    for a, b in zip(wg, we):
        assert np.abs(a - b).max() <= 2e-6 + 1e-5 * np.abs(b).max(), (kind, np.abs(a - b).max())
    for a, b in zip(sg["m"], se["m"]):
        assert np.allclose(a, b, rtol=1e-3, atol=1e-7)
    for a, b in zip(sg["v"], se["v"]):
        assert np.allclose(a, b, rtol=2e-3, atol=1e-10)
    print(f"{kind}: 15 steps {fast} {tg * 1e3:.1f} ms, eager {te * 1e3:.1f} ms")


@pytest.mark.gpu
def test_captured_training_learns_and_feeds_the_engine():
    """The product path end to end on the GPU: CNN.train (captured steps, dropout on) lowers the loss on a learnable
    target, the new weights reach the scoring engine, and a deep copy of the model trains on without the original's
    device state."""
    rng = np.random.default_rng(0)
    alphabet, L = "TGCA", 8
    seqs = ["".join(alphabet[i] for i in row) for row in rng.integers(0, 4, (600, L))]
    y = np.array([s.count("G") / L + 0.5 * (s[0] == "T") for s in seqs], np.float32)
    model = bm.CNN(L, 32, 100, alphabet, seed=4)
    before = float(np.mean((model.get_fitness(seqs) - y) ** 2))
    model.train(seqs, y)
    after = float(np.mean((model.get_fitness(seqs) - y) ** 2))
    assert after < 0.25 * before, (before, after)
    assert model.model._opt_state["t"] == 20 * 3
    twin = copy.deepcopy(model)
    twin.train(seqs, y)
    assert twin.model._opt_state["t"] == 120 and model.model._opt_state["t"] == 60
    assert float(np.mean((twin.get_fitness(seqs) - y) ** 2)) < after * 1.5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["native", "graph"])
def test_ensemble_members_train_side_by_side(mode, monkeypatch):
    """`Ensemble.train` trains its members together (training.fit_many): ONE fx_train_fit call whose launches cover every
    member (native), or the members' captured steps interleaved on one stream per member (graph).  With the members'
    shuffles seeded, the result is what training them one after the other gives -- same bits -- for members of different
    architectures, batch sizes and epoch counts; and through the plugin API every member of a 3-CNN ensemble ends
    with its own weights, step count and a lower loss."""
    import flexs_amd

    monkeypatch.setenv("FLEXS_AMD_TRAIN", mode)

    L, alphabet, n = 9, "UGCA", 700
    seqs, _, y = _batch("mlp", L, alphabet, n, 11)
    specs = [("mlp", 256, 3), ("ge", 128, 2), ("mlp", 256, 3), ("ge", 256, 4)]

    def members():
        return [_model(kind, L, alphabet, 20 + i).model for i, (kind, _, _) in enumerate(specs)]

    together, one_by_one = members(), members()
    training.fit_many(together, seqs, y, [alphabet] * 4, [b for _, b, _ in specs], [e for _, _, e in specs], seeds=[5, 6, 7, 8])
    for k, (arch, (_, b, e)) in enumerate(zip(one_by_one, specs)):
        training.fit(arch, seqs, y, alphabet, batch_size=b, epochs=e, seed=5 + k)
    for a, b_ in zip(together, one_by_one):
        assert a._opt_state["t"] == b_._opt_state["t"]
        assert all(np.array_equal(u, v) for u, v in zip(a.get_weights(), b_.get_weights()))
        assert all(np.array_equal(u, v) for u, v in zip(a._opt_state["v"], b_._opt_state["v"]))

    rng = np.random.default_rng(0)
    seqs8 = ["".join("TGCA"[i] for i in row) for row in rng.integers(0, 4, (500, 8))]
    y8 = np.array([s.count("G") / 8 for s in seqs8], np.float32)
    ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
    before = [float(np.mean((m.get_fitness(seqs8) - y8) ** 2)) for m in ens.models]
    ens.train(seqs8, y8)
    for m, b0 in zip(ens.models, before):
        assert m.model._opt_state["t"] == 20 * 2
        assert float(np.mean((m.get_fitness(seqs8) - y8) ** 2)) < 0.5 * b0
    w0, w1 = ens.models[0].model.get_weights(), ens.models[1].model.get_weights()
    assert not np.array_equal(w0[0], w1[0])
    assert np.array_equal(ens.get_fitness(seqs8[:50]), np.mean(np.stack([m.get_fitness(seqs8[:50]) for m in ens.models], axis=1), axis=1))


def test_ensemble_train_reaches_every_member_on_any_backend():
    """`Ensemble.train` / `AdaptiveEnsemble.train` go through ensemble.train_members: stock device surrogates are handed to
    training.fit_many (on a machine without a GPU that is the one-by-one loop), any other member list -- a member with
    its own `train`, a foreign model -- keeps the reference's loop (ensemble.py:42-52)."""
    import flexs_amd
    from flexs_amd.baselines.models.adaptive_ensemble import AdaptiveEnsemble

    seqs, _, y = _batch("mlp", 8, "TGCA", 40, 3)
    ens = flexs_amd.Ensemble([bm.MLP(8, 16, "TGCA", seed=i, epochs=2, batch_size=16) for i in range(3)])
    ens.train(seqs, y)
    assert [m.model._opt_state["t"] for m in ens.models] == [2 * 3] * 3

    class Recorder(flexs_amd.Model):
        def __init__(self):
            super().__init__("rec")
            self.calls = []

        def train(self, sequences, labels):
            self.calls.append(len(sequences))

        def _fitness_function(self, sequences):
            return np.zeros(len(sequences))

    class OwnTrain(bm.MLP):
        def train(self, sequences, labels, verbose=False):
            self.seen = len(sequences)

    rec, own, stock = Recorder(), OwnTrain(8, 16, "TGCA", seed=5), bm.MLP(8, 16, "TGCA", seed=6, epochs=1)
    flexs_amd.Ensemble([rec, own, stock]).train(seqs, y)
    assert rec.calls == [40] and own.seen == 40 and stock.model._opt_state["t"] == 1
    ada = AdaptiveEnsemble([bm.MLP(8, 16, "TGCA", seed=7, epochs=1), bm.MLP(8, 16, "TGCA", seed=8, epochs=1)])
    ada.train(seqs[:8], y[:8])                               # < 10 samples: no hold-out scoring (adaptive_ensemble.py:84-88)
    assert all(m.model._opt_state["t"] == 1 for m in ada.models)
