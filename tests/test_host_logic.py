"""Host-side logic of the drop-in API (no GPU): names, cost accounting, input
marshalling, error types, constructors -- the scenarios of the reference's own
tests/test_models.py plus the golden Ensemble / AdaptiveEnsemble fixtures on the
foreign-member (host stacking) path."""
import json
import os

import numpy as np
import pytest

import flexs_amd
from flexs_amd import _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils

rng = np.random.default_rng(0)


class FakeModel(flexs_amd.Model):
    def _fitness_function(self, sequences):
        return rng.random(size=len(sequences))

    def train(self, *args, **kwargs):
        pass


class FakeConstantModel(flexs_amd.Model):
    def __init__(self, constant):
        super().__init__(name="ConstantModel")
        self.constant = constant

    def _fitness_function(self, sequences):
        return np.ones(len(sequences)) * self.constant

    def train(self, *args, **kwargs):
        pass


class FixedModel(flexs_amd.Model):
    def __init__(self, name, values):
        super().__init__(name)
        self.values = values
        self.trained = 0

    def _fitness_function(self, sequences):
        return self.values[: len(sequences)]

    def train(self, sequences, labels):
        self.trained += 1


def test_landscape_cost_accounting():
    m = FakeConstantModel(3)
    assert m.cost == 0 and m.name == "ConstantModel"
    m.get_fitness(["A", "B"])
    m.get_fitness(np.array(["A"]))
    assert m.cost == 3
    m.cost = 0                                     # explorers reset it (flexs/explorer.py:126)
    assert m.cost == 0
    lam = flexs_amd.LandscapeAsModel(m)
    assert lam.name == "LandscapeAsModel=ConstantModel"
    assert lam.get_fitness(["A"]).tolist() == [3.0] and lam.cost == 1 and m.cost == 0
    with pytest.raises(TypeError):
        flexs_amd.Landscape("abstract")            # abstract base, like the reference


def test_adaptive_ensemble_reference_scenario():
    """tests/test_models.py:36-52 of the reference."""
    ens = bm.AdaptiveEnsemble([FakeConstantModel(1), FakeConstantModel(2)])
    assert np.sum(ens.weights) == 1
    assert ens.get_fitness(["ATC"]) == 1.5
    assert ens.name == "AdaptiveEns(ConstantModel|ConstantModel)"
    members = [FakeModel(name="FakeModel") for _ in range(2)]
    ens = bm.AdaptiveEnsemble(members)
    ens.train(["ATC"] * 15, list(range(15)))
    assert np.any(ens.weights != np.ones(2) / 2)
    assert np.isclose(np.sum(ens.weights), 1)


def test_ensemble_golden_host_path(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "ensemble.json")))
    arrs = np.load(os.path.join(golden_dir, "ensemble.npz"))
    seqs = meta["sequences"]
    for ci, case in enumerate(meta["cases"]):
        vals = arrs[f"in{ci}"]
        members = [FixedModel(f"m{j}", np.ascontiguousarray(vals[:, j])) for j in range(case["M"])]
        e = flexs_amd.Ensemble(members)
        out = e.get_fitness(seqs)
        assert e.name == case["name"]
        assert out.dtype == np.dtype(case["out_dtype"]) and np.array_equal(out, arrs[f"out{ci}"])
        e.get_fitness(seqs[:10])
        ident = flexs_amd.Ensemble(members, combine_with=lambda x: x).get_fitness(seqs)
        assert np.array_equal(ident, arrs[f"ident{ci}"])
        e.train(seqs, np.zeros(len(seqs)))
        assert e.cost == case["ens_cost"]
        assert [m.cost for m in members] == case["member_costs"]
        assert [m.trained for m in members] == case["member_trained"]
    members = [FixedModel(f"a{j}", np.ascontiguousarray(arrs["ada_in"][:, j])) for j in range(4)]
    ae = bm.AdaptiveEnsemble(members)
    assert ae.name == meta["adaptive"]["name"]
    assert np.array_equal(ae.get_fitness(seqs), arrs["ada_out_default"])
    ae.weights = bm.adaptive_ensemble.r2_weights(arrs["ada_preds"], arrs["ada_labels"])
    assert np.allclose(ae.weights, arrs["ada_w"], rtol=1e-12)
    ae.weights = arrs["ada_w"]
    assert np.array_equal(ae.get_fitness(seqs), arrs["ada_out_w"])
    assert ae.cost == meta["adaptive"]["cost"] and [m.cost for m in members] == meta["adaptive"]["member_costs"]


def test_model_names_and_shapes():
    cnn = bm.CNN(8, 32, 100, "TGCA")
    assert cnn.name == "CNN_hidden_size_100_num_filters_32"            # cnn.py:58-59
    assert cnn.model.count_params() == 22429
    assert bm.CNN(237, 32, 100, s_utils.AAS).model.count_params() == 41373
    mlp = bm.MLP(14, 100, "UGCA")
    assert mlp.name == "MLP_hidden_size_100" and mlp.model.count_params() == 26001
    ge = bm.GlobalEpistasisModel(90, 100, s_utils.AAS)
    assert ge.name == "MLP_hidden_size_100"                            # sic, global_epistasis_model.py:39-40
    assert ge.model.count_params() == 12202
    assert bm.CNN(8, 32, 100, "TGCA", name="x").name == "x"
    assert cnn.batch_size == 256 and cnn.epochs == 20 and cnn.alphabet == "TGCA"
    ens = flexs_amd.Ensemble([cnn, mlp])
    assert ens.name == "Ens(CNN_hidden_size_100_num_filters_32|MLP_hidden_size_100)"
    # Keras-like weight access
    w = cnn.model.get_weights()
    assert [a.shape for a in w][:4] == [(5, 4, 32), (32,), (5, 32, 32), (32,)]
    assert all(np.all(a == 0) for a in w[1::2])                        # zero biases, glorot kernels
    lim = np.sqrt(6.0 / (5 * 4 + 5 * 32))
    assert np.abs(w[0]).max() <= lim and np.abs(w[0]).max() > 0.5 * lim
    with pytest.raises(ValueError):
        cnn.model.set_weights(w[:-1])
    nam = bm.NoisyAbstractModel(FakeConstantModel(2), signal_strength=0.75)
    assert nam.name == "NAMb_ss0.75" and nam.ss == 0.75 and nam.cache == {}


def test_valid_conv_shorter_than_kernel_raises_at_construction():
    with pytest.raises(ValueError):
        bm.CNN(seq_len=3, num_filters=1, hidden_size=1, alphabet="TGCA")   # default kernel_size=5
    bm.CNN(seq_len=3, num_filters=1, hidden_size=1, kernel_size=2, alphabet="TGCA")  # tests/test_models.py:56-62


def test_sequences_to_bytes():
    b = _native.sequences_to_bytes(["ACGT", "TTTT"])
    assert b.dtype == np.uint8 and b.shape == (2, 4) and bytes(b[0]) == b"ACGT"
    assert np.array_equal(_native.sequences_to_bytes(np.array(["ACGT", "TTTT"])), b)
    assert np.array_equal(_native.sequences_to_bytes(np.array([b"ACGT", b"TTTT"])), b)
    assert np.array_equal(_native.sequences_to_bytes(("ACGT", "TTTT")), b)
    assert np.array_equal(_native.sequences_to_bytes([np.str_("ACGT"), np.str_("TTTT")]), b)
    assert _native.sequences_to_bytes([], L=8).shape == (0, 8)
    for ragged in (["ACGT", "TTT"], np.array(["ACGT", "TT"]), ["ACGTA", "TTT"]):
        with pytest.raises(ValueError):
            _native.sequences_to_bytes(ragged)
    with pytest.raises(ValueError):
        _native.sequences_to_bytes(["ACGT"], L=8)
    with pytest.raises(ValueError):
        _native.sequences_to_bytes(["ACሴT"])


def test_lut_first_occurrence():
    lut = _native.make_lut("TGCA")
    assert [lut[ord(c)] for c in "TGCA"] == [0, 1, 2, 3] and lut[ord("X")] == 0xFF
    assert _native.make_lut("ABAC")[ord("A")] == 0     # str.index semantics


def test_sequence_utils_host_helpers():
    assert s_utils.AAS == "ILVAGMFYWEDQNHCRKSTP" and s_utils.RNAA == "UGCA" and s_utils.DNAA == "TGCA" and s_utils.BA == "01"
    muts = s_utils.generate_single_mutants("AT", "TGCA")
    assert muts[0] == "AT" and len(muts) == 1 + 2 * 4 and muts[1] == "TT" and muts[4] == "AT" and muts[5] == "AT"
    seqs = s_utils.generate_random_sequences(7, 5, "TGCA")
    assert len(seqs) == 5 and all(len(s) == 7 and set(s) <= set("TGCA") for s in seqs)
    assert s_utils.generate_random_mutant("ACGT", 0.0, "TGCA") == "ACGT"
    m = s_utils.generate_random_mutant("ACGT" * 10, 1.0, "T")
    assert m == "T" * 40
    base = np.eye(4)[[0, 1, 2, 3]]
    sample = np.zeros((4, 4)); sample[1, 3] = 1
    out = s_utils.construct_mutant_from_sample(sample, base)
    assert out[1].tolist() == [0, 0, 0, 1] and np.array_equal(out[[0, 2, 3]], base[[0, 2, 3]])
    assert s_utils.string_to_one_hot("", "TGCA").shape == (0, 4)


def test_training_reduces_loss_and_bumps_version():
    """PyTorch replacement of model.fit (keras_model.py:49-67): statistical check only."""
    import torch

    torch.manual_seed(0)
    r = np.random.default_rng(1)
    alphabet, L = "TGCA", 8
    seqs = ["".join(alphabet[i] for i in row) for row in r.integers(0, 4, (400, L))]
    labels = np.array([s.count("A") / L for s in seqs])
    from oracle import ref_np    # checker only

    for model in (bm.MLP(L, 32, alphabet, seed=0), bm.GlobalEpistasisModel(L, 16, alphabet, seed=0),
                  bm.CNN(L, 8, 16, alphabet, seed=0)):
        kind = model.model.kind
        before = np.mean((ref_np.keras_fitness(seqs, alphabet, kind, model.model.get_weights()) - labels) ** 2)
        v0 = getattr(model.model, "_version", 0)
        model.train(seqs, labels)
        after = np.mean((ref_np.keras_fitness(seqs, alphabet, kind, model.model.get_weights()) - labels) ** 2)
        assert after < 0.5 * before, (kind, before, after)
        assert model.model._version == v0 + 1
    with pytest.raises(ValueError):
        bm.MLP(L, 8, alphabet).train(["ACGTXCGT"], [0.0])
    with pytest.raises(ValueError):
        bm.MLP(L, 8, alphabet).train(["ACG"], [0.0])


def test_dyna_ppo_ensemble_gating():
    """dyna_ppo.py:118-130: members below the r^2 threshold are ignored; none passing -> best single."""
    a, b, c = FakeConstantModel(1.0), FakeConstantModel(3.0), FakeConstantModel(8.0)
    ens = bm.DynaPPOEnsemble(3, "TGCA", models=[a, b, c])
    assert ens.name == "DynaPPOEnsemble" and list(ens.r_squared_vals) == [1, 1, 1]
    assert ens.get_fitness(["ATC", "ATG"]).tolist() == [4.0, 4.0]
    ens.r_squared_vals = [0.9, 0.1, 0.6]
    assert ens.get_fitness(["ATC"]).tolist() == [4.5]
    ens.r_squared_vals = [0.2, 0.4, 0.1]
    assert ens.get_fitness(["ATC"]).tolist() == [3.0]
    assert (a.cost, b.cost, c.cost) == (3, 3, 3) and ens.cost == 4
    ens.train(["ATC"] * 5, [0.0] * 5)                      # < 10 sequences: no-op (dyna_ppo.py:94-95)
    assert list(ens.r_squared_vals) == [0.2, 0.4, 0.1]
    ens.train(["ATC"] * 12, list(range(12)))               # constant predictions -> r^2 = 0 for every member
    assert list(ens.r_squared_vals) == [0, 0, 0]
    default = bm.DynaPPOEnsemble(14, "UGCA")
    assert [m.name for m in default.models] == ["MLP_hidden_size_100", "MLP_hidden_size_200",
                                                "CNN_hidden_size_100_num_filters_32"]


def test_tf_binding_host_tables_match_reference(golden_dir):
    """Host half of TFBinding (parse + normalise, tf_binding.py:31-41) against values captured from
    the reference class; needs the reference's data file, so it only runs in the build container."""
    path = "/root/reference/flexs/landscapes/data/tf_binding/SIX6_REF_R1_8mers.txt"
    if not os.path.exists(path):
        pytest.skip("reference data files are not available on this machine")
    from flexs_amd.landscapes import TFBinding

    fx = json.load(open(os.path.join(golden_dir, "tf_binding.json")))
    land = TFBinding(path)
    assert land.name == fx["name"] and len(land.sequences) == 65536
    assert [land.sequences[s] for s in fx["sample_sequences"]] == fx["sample_values"]
    assert land.sequences["ATTATGTT"] == fx["tutorial_known_answer"]["value"]
    assert getattr(land, "batch_safe") is True


def test_bench_report_contract():
    """bench.py's JSON line: the keys and types the driver reads (no GPU needed for the assembly itself)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.make_report(world=2, N=100_000, steps=200, warmup=20, elapsed=0.05, host_issue_s=0.004, kern_ms=0.2, use_dist=True)
    json.loads(json.dumps(r))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 2 and r["steps"] == 200 and r["warmup"] == 20 and r["vs_baseline"] is None
    assert r["value"] == pytest.approx(2 * 100_000 * 200 / 0.05) and r["ms_per_step"] == pytest.approx(0.25)
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert rf["flop_per_launch"] == 2.0 * 48628 * 3 * 100_000            # 97 256 FLOP per sequence-member
    assert rf["achieved_algorithmic"] == pytest.approx(rf["flop_per_launch"] / 0.2e-3 / 1e12) and rf["frac_algorithmic"] == pytest.approx(rf["achieved_algorithmic"] / 157.3)
    assert rf["achieved"] == pytest.approx(rf["issued_flop_per_launch"] / 0.2e-3 / 1e12) and rf["frac"] == pytest.approx(rf["achieved"] / 157.3) and rf["frac"] <= 1.0
    assert rf["algorithmic_bytes_per_launch"] == (8 + 12) * 100_000
    # issued-MFMA fraction: 615 MFMAs per 16-sequence tile per member (the kernel's loop bounds), 2048 FLOP each
    assert rf["mfma_per_tile"] == 615 and rf["issued_flop_per_launch"] == 615 * 6250 * 3 * 2048
    assert rf["frac_issued"] == pytest.approx(rf["issued_flop_per_launch"] / 0.2e-3 / 1e12 / 157.3) and rf["frac_issued"] == rf["frac"] < rf["frac_algorithmic"]
    m = bench.make_report(world=8, N=100_000, steps=10, warmup=1, elapsed=0.01, host_issue_s=0.001, kern_ms=0.07, use_dist=True,
                          mode="member", members=8)
    assert m["scaling"] == "strong" and m["value"] == pytest.approx(100_000 * 10 / 0.01) and m["config"]["global_batch"] == 100_000
    assert m["roofline"]["flop_per_launch"] == 2.0 * 48628 * 1 * 100_000      # one member per rank


def test_issued_mfma_counts_follow_the_kernels_loop_bounds():
    """fx_debug_mfma_per_tile against a direct count: conv taps inside the 'same' padding are not issued, the one-hot
    first layers are gathers, the hidden tail tile runs ceil(tail / 4) of its k-steps."""
    from flexs_amd import _native

    def conv_taps(L1, k):
        pl = (k - 1) // 2
        return sum(0 <= t + j - pl < L1 for t in range(L1) for j in range(k))

    for L, A, F, H, K in ((8, 4, 32, 100, 5), (14, 4, 32, 100, 5), (237, 20, 32, 100, 5), (14, 4, 32, 100, 3), (9, 4, 16, 100, 5)):
        ft, ht, tail = -(-F // 16), -(-H // 16), H - 16 * (-(-H // 16) - 1)
        hh = ht * (4 * (ht - 1) + (-(-tail // 4) if tail >= 4 else 1))
        want = (conv_taps(L - K + 1, K) + conv_taps(L - K + 1, A - 1)) * ft * ft * 4 + ft * 4 * ht + hh
        assert _native.mfma_per_tile(_native.FX_CNN, L, A, F, H, K) == want
    assert _native.mfma_per_tile(_native.FX_CNN, 8, 4, 32, 100, 5) == 615      # PMC-confirmed in round 1 (r1_run35)
    assert _native.mfma_per_tile(_native.FX_MLP, 14, 4, 0, 100, 0) == 2 * 175
    assert _native.mfma_per_tile(_native.FX_GE, 90, 20, 0, 100, 0) == 175
    assert _native.mfma_per_tile(_native.FX_MLP, 14, 4, 0, 64, 0) == 2 * 4 * 16      # H = 64: four full tiles
    assert _native.mfma_per_tile(_native.FX_MLP, 14, 4, 0, 80, 0) == 2 * 7 * 28      # 5 real tiles rounded up to the 7-tile kernel


def test_ragged_rows_for_edit_distance():
    """Row format of the edit-distance entry points: NUL-padded, strict about length and byte range."""
    from flexs_amd import _native

    rows = _native.ragged_to_bytes(["ACG", "", "ACGTAC"], 6)
    assert rows.dtype == np.uint8 and rows.tolist() == [[65, 67, 71, 0, 0, 0], [0] * 6, [65, 67, 71, 84, 65, 67]]
    assert _native.ragged_to_bytes([], 4).shape == (0, 4)
    for bad in (["ACGTACG"], ["A\x00C"], ["A\u0394"]):
        with pytest.raises(ValueError):
            _native.ragged_to_bytes(bad, 6)


def test_string_marshalling_helper_equals_pure_python(monkeypatch):
    """csrc/strpack.c (CPython helper behind get_fitness(list[str])) and the pure-Python path: same bytes,
    same exceptions."""
    from flexs_amd import _native

    assert _native._strpack is not None, "flexs_amd/_strpack*.so was not built (make -C flexs_amd/csrc)"
    rng = np.random.default_rng(0)
    seqs = ["".join("ILVAGMFYWEDQNHCRKSTP"[i] for i in r) for r in rng.integers(0, 20, (500, 31))]
    cases = [seqs, tuple(seqs[:7]), [np.str_(s) for s in seqs[:5]], ["AC\xe9", "TT\xff"], [""], ["", ""]]
    bad = [(["ACG", "AC"], ValueError), (["AC", "ACG"], ValueError), (["AC\u0394", "ACG"], ValueError), (["ACG", 5], TypeError)]
    got = [_native.sequences_to_bytes(c) for c in cases]
    for c, exc in bad:
        with pytest.raises(exc):
            _native.sequences_to_bytes(c)
    monkeypatch.setattr(_native, "_strpack", None)
    for c, g in zip(cases, got):
        assert np.array_equal(_native.sequences_to_bytes(c), g) and g.dtype == np.uint8
    for c, exc in bad:
        with pytest.raises(exc):
            _native.sequences_to_bytes(c)


def test_models_copy_and_pickle_without_device_handles():
    """copy.deepcopy / pickle of the host-side objects must not try to duplicate device handles: they are dropped
    and rebuilt lazily by the copy."""
    import copy
    import pickle

    from flexs_amd.baselines.models.noisy_abstract_model import NoisyAbstractModel

    cnn = bm.CNN(8, 32, 100, "TGCA", seed=3)
    cnn.cost = 7
    cnn._native_model, cnn._native_version = object(), (1, 2)          # stand-ins for a live fx_model
    state = cnn.__getstate__()
    assert state["_native_model"] is None and state["_native_version"] is None and state["cost"] == 7
    assert all(np.array_equal(a, b) for a, b in zip(state["model"].get_weights(), cnn.model.get_weights()))
    cnn._native_model = None
    twin = pickle.loads(pickle.dumps(cnn))
    assert twin.name == cnn.name and twin.cost == 7 and twin._native_model is None
    assert all(np.array_equal(a, b) for a, b in zip(twin.model.get_weights(), cnn.model.get_weights()))
    ens = copy.deepcopy(flexs_amd.Ensemble([cnn, bm.MLP(8, 16, "TGCA", seed=1)]))
    assert [m.name for m in ens.models] == [cnn.name, "MLP_hidden_size_16"]

    class Flat(flexs_amd.Landscape):
        def __init__(self):
            super().__init__("flat")

        def _fitness_function(self, seqs):
            return np.zeros(len(seqs))

    nam = NoisyAbstractModel(Flat(), 0.5)
    nam.cache = {"ACGT": 1.0}
    nam._dev_cache, nam._dev_keys = object(), ["ACGT"]
    st = nam.__getstate__()
    assert st["_dev_cache"] is None and st["_dev_keys"] == [] and st["cache"] == {"ACGT": 1.0}


def test_sequence_generators_match_reference(golden_dir):
    """generate_single_mutants / generate_random_sequences / generate_random_mutant / construct_mutant_from_sample
    against outputs of the reference's own functions under fixed `random` seeds (same values, same RNG position)."""
    import random

    g = json.load(open(os.path.join(golden_dir, "sequence_generators.json")))
    for c in g["single_mutants"]:
        assert s_utils.generate_single_mutants(c["wt"], c["alphabet"]) == c["out"]
    for c in g["random_sequences"]:
        random.seed(c["seed"])
        assert s_utils.generate_random_sequences(c["length"], c["number"], c["alphabet"]) == c["out"]
        assert random.random() == c["next_random"]
    for c in g["random_mutant"]:
        random.seed(c["seed"])
        assert [s_utils.generate_random_mutant(c["sequence"], c["mu"], c["alphabet"]) for _ in range(4)] == c["out"]
        assert random.random() == c["next_random"]
    for c in g["construct_mutant"]:
        out = s_utils.construct_mutant_from_sample(np.array(c["sample"]), np.array(c["base"]))
        assert out.tolist() == c["out"] and str(out.dtype) == c["dtype"]


def test_string_packing_with_worker_threads():
    """csrc/strpack.c packs big batches with a persistent pool of worker threads (GIL released; the workers only read
    immutable str objects): same bytes as one thread at every thread count, sub-ranges included, and the FIRST bad item
    in row order decides the status -- whichever share it falls in."""
    from flexs_amd import synth

    sp = _native._strpack
    assert sp is not None
    L, N = 24, 40_000                                    # 960 kB: above the threshold for the pool
    b = synth.random_sequence_bytes(N, L, "ILVAGMFYWEDQNHCRKSTP", 3)
    seqs = synth.bytes_to_strings(b)
    prev = sp.set_threads(1)
    try:
        for threads in (1, 2, 3, 5, 8, 16, 0):
            sp.set_threads(threads)
            out = np.zeros((N, L), np.uint8)
            assert sp.pack(seqs, L, out) == 0 and np.array_equal(out, b)
            part = np.zeros((30_001, L), np.uint8)
            assert sp.pack(seqs, L, part, 7_777, 30_001) == 0 and np.array_equal(part, b[7_777:37_778])
        sp.set_threads(4)
        out = np.zeros((N, L), np.uint8)
        for where, bad, status in ((35_000, "A" * (L - 1), 1), (21_000, "A" * (L - 1) + "Δ", 2), (9_000, 7, 3)):
            seqs[where] = bad                            # later shares fail first in time, earlier rows win
            assert sp.pack(seqs, L, out) == status
        seqs[5] = "x" * (L + 1)
        assert sp.pack(seqs, L, out) == 1
        # wide (UCS-2) strings that still fit one byte per character go through the per-character path in the workers too
        wide = [("Δ" + "A" * L)[1:] for _ in range(N)]
        assert sp.pack(wide, L, out) == 0 and (out == ord("A")).all()
    finally:
        sp.set_threads(prev)


def test_string_packing_from_two_python_threads_at_once():
    """The packing pool releases the GIL while it runs, so a second Python thread can enter `pack` meanwhile: it must not
    disturb the job in flight (it packs on its own thread) and both results must be right."""
    import threading

    from flexs_amd import synth

    sp = _native._strpack
    prev = sp.set_threads(4)
    try:
        L, N = 32, 60_000
        jobs = []
        for seed in (1, 2, 3, 4):
            b = synth.random_sequence_bytes(N, L, "UGCA", seed)
            jobs.append((b, synth.bytes_to_strings(b), np.zeros((N, L), np.uint8)))
        errors = []

        def run(b, seqs, out):
            for _ in range(6):
                out[:] = 0
                if sp.pack(seqs, L, out) != 0 or not np.array_equal(out, b):
                    errors.append("mismatch")

        threads = [threading.Thread(target=run, args=j) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
    finally:
        sp.set_threads(prev)


def test_score_small_packs_and_calls_through_the_plan():
    """csrc/strpack.c score_small: the explorer-size fast path packs the strings and calls the function the plan carries
    (fx_score in production; a ctypes callback here, so the argument passing is checked without a GPU)."""
    import ctypes as C
    import struct

    from flexs_amd import _native

    if not _native._HAS_SCORE_SMALL:
        pytest.skip("strpack helper not built")
    seen = []
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint8), C.c_longlong, C.c_int,
                     C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))

    def fake_fx_score(e, models, M, ascii, N, L, lut, out_nm, out_mean):
        seen.append((e, [models[i] for i in range(M)], N, L, bytes(ascii[:N * L]), bool(out_nm), bool(out_mean), bytes(lut[:256])))
        o = out_nm if out_nm else out_mean
        for i in range(N * (M if out_nm else 1)):
            o[i] = i + 0.5
        return -3 if N == 3 else 0

    fn = FN(fake_fx_score)
    lut = _native.make_lut("ACGT")
    for want in (1, 2):
        plan = struct.pack("PPqqq16P256sPPPqq", C.cast(fn, C.c_void_p).value, 777, 2, 4, want, *([11, 22] + [0] * 14), lut.tobytes(),
                           0, 0, 0, 0, 0)                              # (no streamed calls)
        out = np.empty((2, 2) if want == 1 else (2,), np.float32)
        assert _native._strpack.score_small(plan, ["ACGT", "TTTT"], out) == 0
        assert seen[-1] == (777, [11, 22], 2, 4, b"ACGTTTTT", want == 1, want == 2, lut.tobytes())
        assert np.array_equal(out.ravel(), np.arange(out.size) + 0.5)
        assert _native._strpack.score_small(plan, ("ACGT", "TTTT"), out) == 0
    assert _native._strpack.score_small(plan, ["ACGT", "TTT"], out) == 1001           # ragged
    assert _native._strpack.score_small(plan, ["ACGT", "TTTሴ"], out) == 1002      # not latin-1
    assert _native._strpack.score_small(plan, ["ACGT", 5], out) == 1003               # not a str
    assert _native._strpack.score_small(plan, ["ACGT"] * 3, np.empty(3, np.float32)) == 2003   # the callee's FX_EBADCHAR
    assert _native._strpack.score_small(plan, ["ACGT"] * 16385, np.empty(16385, np.float32)) == -1  # more than 64 KiB of sequence bytes: too big for this path
    assert _native._strpack.score_small(plan, ["ACGT", "TTTT"], np.empty(1, np.float32)) == -1     # output too small
    assert _native._strpack.score_small(b"xx", ["ACGT"], out) == -1
    assert _native._strpack.score_small(plan, np.array(["ACGT"]), out) == -1          # not a list / tuple
    n_calls = len(seen)
    assert _native._strpack.score_small(plan, [], out) == -1 and len(seen) == n_calls


def test_score_small_streams_through_the_plan():
    """csrc/strpack.c score_small, streamed form (include/flexs_amd.h fx_score_stream_*): calls of at least stream_min strings
    are packed piece by piece into the area _begin hands out, every piece but the last reported with _rows, the answers come
    from _end; a refusal from _begin or an FX_EUNSUPPORTED from _end lands in the packed call; a string that cannot be packed
    closes the stream with ok = 0.  Fake callbacks: the protocol is checked without a GPU."""
    import ctypes as C
    import struct

    from flexs_amd import _native

    if not _native._HAS_SCORE_SMALL:
        pytest.skip("strpack helper not built")
    log = []
    area = (C.c_uint8 * 4096)()
    behaviour = {"begin": 0, "end": 0}
    SCORE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint8), C.c_longlong, C.c_int,
                        C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))
    BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_uint8),
                        C.POINTER(C.POINTER(C.c_uint8)))
    ROWS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong)
    END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

    def fake_score(e, models, M, ascii, N, L, lut, out_nm, out_mean):
        log.append(("score", N, bytes(ascii[:N * L])))
        for i in range(N):
            out_mean[i] = -1.0
        return 0

    def fake_begin(e, models, M, N, L, lut, rows):
        log.append(("begin", e, M, N, L))
        if behaviour["begin"]:
            return behaviour["begin"]
        rows[0] = C.cast(area, C.POINTER(C.c_uint8))
        return 0

    def fake_rows(e, n):
        log.append(("rows", n, bytes(area[:n * 4])))
        return 0

    def fake_end(e, ok, out_nm, out_mean):
        log.append(("end", ok))
        if ok and not behaviour["end"]:
            for i in range(10):
                out_mean[i] = i + 0.25
        return behaviour["end"] if ok else 0

    fns = [SCORE(fake_score), BEGIN(fake_begin), ROWS(fake_rows), END(fake_end)]
    ptr = [C.cast(f, C.c_void_p).value for f in fns]
    lut = _native.make_lut("ACGT")
    plan = struct.pack("PPqqq16P256sPPPqq", ptr[0], 777, 1, 4, 2, *([11] + [0] * 15), lut.tobytes(), ptr[1], ptr[2], ptr[3], 8, 4)
    seqs = ["ACGT", "CCCC", "GGGG", "TTTT", "AAAA", "ACAC", "GTGT", "TGCA", "CATG", "GGCC"]
    packed = "".join(seqs).encode()
    out = np.zeros(10, np.float32)
    assert _native._strpack.score_small(plan, seqs, out) == 0
    assert log == [("begin", 777, 1, 10, 4), ("rows", 4, packed[:16]), ("rows", 8, packed[:32]), ("end", 1)]
    assert bytes(area[:40]) == packed and np.array_equal(out, np.arange(10) + 0.25)
    del log[:]
    assert _native._strpack.score_small(plan, seqs[:7], out) == 0                   # fewer than stream_min strings: the packed call
    assert log == [("score", 7, packed[:28])]
    del log[:]
    behaviour["begin"] = -7                                                          # FX_EUNSUPPORTED: no generation for this call
    assert _native._strpack.score_small(plan, seqs, out) == 0
    assert log == [("begin", 777, 1, 10, 4), ("score", 10, packed)] and (out == -1.0).all()
    del log[:]
    behaviour["begin"], behaviour["end"] = 0, -7                                     # the generation went away before it answered
    assert _native._strpack.score_small(plan, seqs, out) == 0
    assert [x[0] for x in log] == ["begin", "rows", "rows", "end", "score"] and log[-1] == ("score", 10, packed)
    del log[:]
    behaviour["end"] = -3                                                            # the callee's FX_EBADCHAR
    assert _native._strpack.score_small(plan, seqs, out) == 2003 and log[-1] == ("end", 1)
    del log[:]
    behaviour["end"] = 0
    assert _native._strpack.score_small(plan, seqs[:9] + ["GGC"], out) == 1001      # ragged, found in the last piece
    assert log[-1] == ("end", 0) and not any(x[0] == "score" for x in log)
    del log[:]
    assert _native._strpack.score_small(plan, seqs[:2] + [5] + seqs[3:], out) == 1003
    assert log == [("begin", 777, 1, 10, 4), ("end", 0)]                            # (found in the first piece: nothing was reported)


@pytest.mark.parametrize("A", [2, 4, 7, 8, 9, 20, 23])
def test_host_argmax_follows_numpys_rule(A):
    """csrc/strpack.c decode_argmax (`one_hot_to_string`, flexs/utils/sequence_utils.py:50-66, for a population on the host): both
    code paths -- four interleaved scalar rows (A < 8) and the 256-bit form -- pick what np.argmax picks: the FIRST maximum, a NaN
    is a maximum (the first NaN wins), +0 / -0 are equal, infinities, constant rows; single-threaded and over the packing threads."""
    from flexs_amd import _native

    sp = _native._strpack
    if sp is None or not hasattr(sp, "decode_argmax"):
        pytest.skip("strpack helper not built")
    alpha = bytes(range(65, 65 + A))
    rng = np.random.default_rng(A)
    x = rng.standard_normal((20000, A))
    x[5] = 0.0
    x[6, A - 1] = np.nan
    x[7, :] = np.nan
    x[8, 0] = -0.0; x[8, 1:] = -1.0
    x[9] = x[9].max()
    x[10, -1] = 1e300
    x[11, A - 2] = np.inf; x[11, A - 1] = np.inf
    x[12] = -np.inf
    x[13, :] = -0.0; x[13, A // 2] = 0.0
    x[14, 0] = np.nan; x[14, 1] = np.inf
    x[100:200] = np.round(x[100:200])                    # many ties
    x[200:300, 1::2] = np.nan                            # NaNs at odd positions: index 1 wins
    want = np.frombuffer(alpha, np.uint8)[np.argmax(x, axis=1)]
    for threads in (1, 0):
        prev = sp.set_threads(threads)
        try:
            out = np.zeros(x.shape[0], np.uint8)
            assert sp.decode_argmax(x, x.shape[0], A, alpha, out) == 0
            assert np.array_equal(out, want), np.nonzero(out != want)[0][:10]
            few = np.zeros(3, np.uint8)                  # fewer rows than one interleaved group
            assert sp.decode_argmax(np.ascontiguousarray(x[5:8]), 3, A, alpha, few) == 0 and np.array_equal(few, want[5:8])
        finally:
            sp.set_threads(prev)
    assert sp.decode_argmax(x, x.shape[0], A, alpha[:A - 1], out) == 1          # alphabet shorter than the rows: refused


def test_adalead_children_in_c_consume_the_random_stream_like_the_python_loop():
    """csrc/strpack.c adalead_children (one tree level of Adalead's roll-outs, adalead.py:128-150): the same children, parents and
    -- after every call -- the same state of Python's `random` generator as the Python loop it replaces, for several alphabets
    (power-of-two and not: `_randbelow`'s redraws), mutation rates, re-draws of children that were seen before; arguments it does
    not handle are handed back (None)."""
    import random

    from flexs_amd import _native
    from flexs_amd.utils import rollouts

    sp = _native._strpack
    if sp is None or not hasattr(sp, "adalead_children"):
        pytest.skip("strpack helper not built")
    assert rollouts._c_children_ok()
    saved = random.getstate()
    try:
        for seed, alphabet, L, mu, n_nodes in ((1, "TGCA", 8, 1, 20), (2, "ILVAGMFYWEDQNHCRKSTP", 40, 2, 13), (3, "ABC", 4, 1, 9),
                                               (4, "AB", 3, 1, 4), (5, "ACGTN", 12, 5, 1), (6, "UGCA", 14, 0.5, 30)):
            rnd = random.Random(seed)
            nodes = [(i, "".join(rnd.choice(alphabet) for _ in range(L))) for i in range(n_nodes)]
            before = {s_ for _, s_ in nodes[::2]}
            now = {s_: 0.0 for _, s_ in nodes[1::3]}
            random.seed(seed)
            want, states = [], []
            for _ in range(5):                               # consecutive levels share the stream
                want.append(rollouts._children_py(nodes, mu, alphabet, before, now))
                states.append(random.getstate())
            random.seed(seed)
            for w, st in zip(want, states):
                got = sp.adalead_children(nodes, mu, alphabet, before, now, random.random, random.getrandbits)
                assert (list(got[0]), list(got[1])) == w and random.getstate() == st
                assert all(c not in before and c not in now for c in got[1])
        nodes = [(0, "ACGT")]
        assert sp.adalead_children(tuple(nodes), 1, "ACGT", set(), {}, random.random, random.getrandbits) is None    # not a list
        assert sp.adalead_children(nodes, 1, "ACGT", [], {}, random.random, random.getrandbits) is None              # not a set
        assert sp.adalead_children([(0, "AC\u1234T")], 1, "ACGT", set(), {}, random.random, random.getrandbits) is None
        with pytest.raises(ZeroDivisionError):
            sp.adalead_children(nodes, 1, "ACGT", set(), {}, lambda: 1 / 0, random.getrandbits)                      # the callable's error comes through
    finally:
        random.setstate(saved)


def test_seen_sequences_bookkeeping_without_a_gpu():
    """flexs_amd.utils.edit_distance.SeenSequences (DyNA-PPO's `all_seqs` + `sequence_density`, environments/dyna_ppo.py:106-114) over
    a stand-in for the device cache (distances by the C oracle): dict semantics of add / add_many (re-adding updates the fitness,
    duplicates inside a batch, ONE key upload per batch), the float64 mirror of the fitness values the C density sums read, the
    batch densities against the reference's per-sequence loop -- with float64 values (C sums) and with a float32 value among them
    (Python operations, as NumPy's scalar rules make the reference compute)."""
    from flexs_amd import _native
    from flexs_amd.utils.edit_distance import SeenSequences
    from oracle import c_oracle

    class FakeCache:
        def __init__(self, L):
            self.L, self.keys, self.appends = L, [], 0
        def __len__(self):
            return len(self.keys)
        def append(self, rows):
            self.appends += 1
            self.keys.extend(bytes(r).rstrip(b"\0") for r in np.asarray(rows))
        def distances(self, q, mode=None):
            q = [bytes(r).rstrip(b"\0") for r in np.asarray(q)]
            return np.array([[min(c_oracle.levenshtein(a, k), 255) for k in self.keys] for a in q], np.uint8).reshape(len(q), len(self.keys))
        def density(self, q, fitness, radius, mode=None):
            d = self.distances(q)
            dens, cnt = np.zeros(len(d)), np.zeros(len(d), np.int32)
            for r, row in enumerate(d):
                for i, di in enumerate(row):
                    if 0 < di <= radius:
                        dens[r] += fitness[i] / float(di); cnt[r] += 1
            return dens, cnt

    def reference_density(all_seqs, seq, radius=2):                       # dyna_ppo.py:106-114
        dens = 0
        for s_ in all_seqs:
            dist = c_oracle.levenshtein(s_.encode(), seq.encode())
            if dist != 0 and dist <= radius:
                dens += all_seqs[s_] / dist
        return dens

    rng = np.random.default_rng(8)
    for use_f32 in (False, True):
        L = 12
        seen = SeenSequences.__new__(SeenSequences)
        seen._L, seen._mode, seen._cache = L, _native.FX_LEVENSHTEIN, FakeCache(L)
        seen._index, seen._fitness = {}, []
        seen._fit_arr, seen._fit_f64 = np.empty(4, np.float64), True      # (small: the mirror grows by doubling)
        all_seqs = {}
        base = "ACGTACGTACGT"
        def mutant():
            s_ = list(base)
            for _ in range(int(rng.integers(0, 3))):
                s_[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
            return "".join(s_)[: int(rng.integers(L - 1, L + 1))]
        for step in range(12):
            batch = [mutant() for _ in range(7)] + ([base] if step % 3 == 0 else [])
            batch[2] = batch[0]                                           # a duplicate inside the batch: the later fitness wins
            fits = [float(v) for v in rng.random(len(batch))]
            if use_f32 and step == 5:
                fits[1] = np.float32(fits[1])
            before = seen._cache.appends
            seen.add_many(batch, fits)
            assert seen._cache.appends - before <= 1                      # one upload (none when nothing was new)
            for s_, f in zip(batch, fits):
                all_seqs[s_] = f
            assert len(seen) == len(all_seqs) == len(seen._cache)
            assert all(seen[s_] == all_seqs[s_] for s_ in all_seqs)
            qs = batch + [mutant() for _ in range(3)]
            assert seen.densities(qs) == [reference_density(all_seqs, q) for q in qs], (use_f32, step)
            assert seen.densities(qs, 1) == [reference_density(all_seqs, q, 1) for q in qs]
        assert (seen._fit_array() is None) == use_f32
        if not use_f32:
            assert np.array_equal(seen._fit_array(), np.array(seen._fitness))
        seen.add(base, 0.125)                                             # the single-sequence form updates in place too
        assert seen[base] == 0.125 and len(seen) == len(all_seqs)
        for call in (lambda: seen.densities([base], 255), lambda: seen.density(base, 300)):
            with pytest.raises(ValueError):                               # (distances come back as min(d, 255) in one byte)
                call()
        assert seen.densities([base], 254) == [reference_density(all_seqs | {base: 0.125}, base, 254)]


def test_population_step_decodes_scores_and_names_in_one_call():
    """csrc/strpack.c population_step (CMA-ES / DyNA-PPO decode-then-score, cmaes.py:61-67, 83-93): per-position argmax of the
    (P, L, A) array by NumPy's rule, those rows handed to the function the plan carries (fx_score in production; a ctypes callback
    here), the rows back as Python strings.  No GPU."""
    import ctypes as C
    import struct

    from flexs_amd import _native

    sp = _native._strpack
    if sp is None or not hasattr(sp, "population_step"):
        pytest.skip("strpack helper not built")
    seen = []
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint8), C.c_longlong, C.c_int,
                     C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))

    def fake_fx_score(e, models, M, ascii, N, L, lut, out_nm, out_mean):
        seen.append((N, L, bytes(ascii[:N * L]), bool(out_nm), bool(out_mean)))
        o = out_nm if out_nm else out_mean
        for i in range(N * (M if out_nm else 1)):
            o[i] = i + 0.25
        return -3 if N == 3 else 0

    fn = FN(fake_fx_score)
    alpha, L, A = "ILVAGMFYWEDQNHCRKSTP", 11, 20
    lut = _native.make_lut(alpha)
    rng = np.random.default_rng(4)
    for want, M in ((2, 3), (1, 1)):
        plan = struct.pack("PPqqq16P256sPPPqq", C.cast(fn, C.c_void_p).value, 777, M, L, want, *([11] * M + [0] * (16 - M)), lut.tobytes(),
                           0, 0, 0, 0, 0)
        for P in (1, 5, 40):
            x = rng.standard_normal((P, L, A))
            x[0, 0, :] = 0.0                                             # a tie: the first letter wins
            x[0, 1, 3] = np.nan                                          # a NaN is a maximum
            codes = np.argmax(x, axis=2)
            want_rows = ["".join(alpha[c] for c in row) for row in codes]
            chars = np.zeros((P, L), np.uint8)
            out = np.zeros((P, M) if want == 1 else (P,), np.float32)
            st, names = sp.population_step(plan, x, P, A, alpha.encode(), chars, out)
            assert st == 0 and names == want_rows
            assert [r.tobytes().decode() for r in chars] == want_rows
            assert seen[-1] == (P, L, "".join(want_rows).encode(), want == 1, want == 2)
            assert np.array_equal(out.ravel(), np.arange(out.size) + 0.25)
    x = rng.standard_normal((3, L, A))
    st, names = sp.population_step(plan, x, 3, A, alpha.encode(), np.zeros((3, L), np.uint8), np.zeros((3, 1), np.float32))
    assert st == 2003 and names is None                                   # the callee's FX_EBADCHAR comes through
    st, names = sp.population_step(plan, x, 3, A, alpha[:5].encode(), np.zeros((3, L), np.uint8), np.zeros((3, 1), np.float32))
    assert st == 1 and names is None                                      # alphabet shorter than the rows
    st, names = sp.population_step(b"xx", x, 3, A, alpha.encode(), np.zeros((3, L), np.uint8), np.zeros((3, 1), np.float32))
    assert st == 1 and names is None


def test_answer_lines_are_collected_like_the_scalar_loop():
    """csrc/host_collect.cc fx_collect_lines (the vector path of the resident form's answer collection): whole 64-byte lines whose
    eight tags equal the request's are unpacked -- scores bit for bit, the bad-character bit reported --, the walk stops at the first
    line with an answer still missing (stale tag, or a tag that differs only in its high bits) and at a tail shorter than a line;
    an unaligned start is handed back to the scalar loop.  Synthetic answer memory: no GPU."""
    import ctypes as C

    from flexs_amd import _native

    lib = _native.lib()
    if not hasattr(lib, "fx_collect_lines"):
        pytest.skip("library without the vector collection")
    fn = lib.fx_collect_lines
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(3)
    N, seq = 100, 0x1234567
    raw = np.zeros(N + 64, np.uint64)                                  # (+ slack: the routine prefetches ahead)
    base = raw.ctypes.data
    off = (-base) % 64 // 8                                            # first 64-byte aligned element
    ans = raw[off:off + N]
    assert ans.ctypes.data % 64 == 0
    scores = rng.standard_normal(N).astype(np.float32)
    scores[5] = np.float32(-0.0); scores[6] = np.float32(3.4e38)
    bits = scores.view(np.uint32).astype(np.uint64)

    def fill(arrived, tag=seq, bad_at=()):
        ans[:] = (np.uint64(tag - 1) << np.uint64(32)) | bits          # stale answers of the previous request
        for i in range(arrived):
            ans[i] = (np.uint64(tag) << np.uint64(32)) | bits[i] | (np.uint64(1 << 63) if i in bad_at else np.uint64(0))

    def collect(n0, n_total):
        out = np.full(N, np.float32(7.0))
        bad = C.c_int(0)
        stop = fn(ans.ctypes.data, out.ctypes.data, n0, n_total, seq, 64, C.byref(bad))
        return stop, out, bad.value

    have_avx2 = True
    fill(N)
    stop, out, bad = collect(0, N)
    if stop == 0:
        have_avx2 = False                                              # (a CPU without AVX2: everything goes to the scalar loop)
    else:
        assert stop == 96 and bad == 0                                 # twelve whole lines; the four-answer tail is the scalar loop's
        assert np.array_equal(out[:96].view(np.uint32), scores[:96].view(np.uint32)) and (out[96:] == 7.0).all()
    if have_avx2:
        fill(43)
        stop, out, bad = collect(0, N)                                  # line 5 (answers 40 .. 47) is not complete
        assert stop == 40 and np.array_equal(out[:40].view(np.uint32), scores[:40].view(np.uint32)) and (out[40:] == 7.0).all()
        stop, out, bad = collect(16, N)
        assert stop == 40 and (out[:16] == 7.0).all() and np.array_equal(out[16:40].view(np.uint32), scores[16:40].view(np.uint32))
        fill(N, bad_at=(9, 70))
        stop, out, bad = collect(0, N)
        assert stop == 96 and bad == 1 and np.array_equal(out[:96].view(np.uint32), scores[:96].view(np.uint32))
        fill(N, bad_at=(99,))                                           # (in the tail: not this routine's to see)
        assert collect(0, N)[2] == 0
        fill(N)
        ans[20] = (np.uint64(seq ^ 0x40000000) << np.uint64(32)) | bits[20]      # a tag that differs in a high bit only
        assert collect(0, N)[0] == 16
        fill(N)
        assert collect(4, N)[0] == 4                                     # not a line start: the scalar loop's
        stop, out, bad = collect(0, 7)
        assert stop == 0                                                 # fewer than eight answers
        # an unaligned answer array is never touched
        shifted = raw[off + 1:off + 1 + 32]
        bad_c = C.c_int(0)
        assert fn(shifted.ctypes.data, out.ctypes.data, 0, 32, seq, 64, C.byref(bad_c)) == 0


def test_epoch_orders_are_uniform_permutations():
    """fx_train_orders (the fit's epoch shuffles): every row is a permutation of 0 .. n - 1, a function of the seed alone, different
    from epoch to epoch and seed to seed; every position receives every row about equally often (a coarse chi-square over many
    shuffles of a short range -- catches a biased bounded draw or an off-by-one in the Fisher-Yates bounds); edge sizes."""
    from flexs_amd import _native

    a = _native.train_orders(12345, 1000, 20)
    assert a.shape == (20, 1000) and a.dtype == np.int32
    assert all(np.array_equal(np.sort(r), np.arange(1000)) for r in a)
    assert np.array_equal(a, _native.train_orders(12345, 1000, 20))
    assert np.array_equal(a[:5], _native.train_orders(12345, 1000, 5))            # (epochs come in sequence from the one stream)
    assert len({r.tobytes() for r in a}) == 20
    assert not np.array_equal(a, _native.train_orders(12346, 1000, 20))
    assert _native.train_orders(1, 0, 3).shape == (3, 0) and _native.train_orders(1, 5, 0).shape == (0, 5)
    assert np.array_equal(_native.train_orders(7, 1, 4), np.zeros((4, 1), np.int32))
    n, reps = 7, 70000
    s = _native.train_orders(99, n, reps)
    counts = np.zeros((n, n))
    for pos in range(n):
        counts[pos] = np.bincount(s[:, pos], minlength=n)
    expected = reps / n
    chi2 = ((counts - expected) ** 2 / expected).sum()                              # 36 degrees of freedom: mean 36, sd 8.5
    assert chi2 < 36 + 6 * 8.5, chi2
    pairs = np.bincount(s[:, 0] * n + s[:, 1], minlength=n * n).reshape(n, n)     # first two positions jointly: n (n - 1) cells
    off = pairs[~np.eye(n, dtype=bool)]
    assert pairs.trace() == 0 and ((off - reps / (n * (n - 1))) ** 2 / (reps / (n * (n - 1)))).sum() < 41 + 6 * 9.1


def test_rng_checkpoint_puts_numpys_global_stream_back():
    """noisy_abstract_model._rng_checkpoint (the fused NoisyAbstractModel batch draws before it knows whether the batch is
    its to answer): after restore() the global legacy RNG is exactly where it was -- also across the 624-word refill --
    and `np.random.exponential(scale=array)` is `scale * standard_exponential(size)` draw for draw."""
    from flexs_amd.baselines.models.noisy_abstract_model import _rng_checkpoint

    def same(a, b):
        return all(np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y for x, y in zip(a, b))

    for seed, burn, draws in ((7, 1000, 37), (8, 310, 500), (9, 0, 1), (10, 623, 2)):
        np.random.seed(seed)
        np.random.random(burn)
        np.random.standard_normal(1)                       # (leaves a cached Gaussian behind: must survive too)
        before = np.random.get_state()
        restore = _rng_checkpoint()
        first = np.random.standard_exponential(draws)
        restore()
        assert same(before, np.random.get_state())
        assert np.array_equal(np.random.standard_exponential(draws), first)
    for seed in range(4):
        scale = np.random.default_rng(seed).random(50) * 3
        scale[3] = 0.0
        np.random.seed(seed)
        a = np.random.exponential(scale=scale)
        pos_a = np.random.get_state()[2]
        np.random.seed(seed)
        b = scale * np.random.standard_exponential(50)
        assert np.array_equal(a, b) and np.random.get_state()[2] == pos_a


def test_rng_checkpoint_self_check_and_both_paths():
    """`_rng_checkpoint` (NoisyAbstractModel's fused query hands a batch back to the one-by-one path with NumPy's global
    stream restored): the raw-state shortcut is only taken after `_mt_state_layout_ok` has confirmed that the bytes at
    `state_address` ARE the MT19937 state; the check itself leaves the user's stream (Gaussian cache included) alone, and
    the documented slow path gives the same restore."""
    from flexs_amd.baselines.models import noisy_abstract_model as nm

    np.random.seed(123)
    np.random.standard_normal(3)                                    # leaves a cached Gaussian in the legacy state
    before = np.random.get_state()
    ok = nm._mt_state_layout_ok()
    after = np.random.get_state()
    assert ok is True                                               # (NumPy 2.2's layout; a False here only means "slow path")
    assert np.array_equal(before[1], after[1]) and before[2:] == after[2:]
    for fast in (True, False):
        nm._MT_FAST = fast
        np.random.seed(7)
        restore = nm._rng_checkpoint()
        a = np.random.exponential(size=700)
        restore()
        assert np.array_equal(a, np.random.exponential(size=700))
    nm._MT_FAST = None


def test_bench_contract_line_is_short_and_carries_every_config():
    """Round-5 verdict item 1: the driver lost a 24.8 KB bench line (BENCH_r05.parsed = null).  The contract line is now built by
    `bench.contract_line` from the full record: <= 4 KB, one line, json round trip, contract keys + `roofline` + `cpu_baseline` + the
    end-to-end figure at the top level, per-config figures as flat scalars of `roofline`; the verbose blocks go to the full record only.
    Checked on the committed full record of round 5 (same block shapes)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(__file__))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r5_bench_driver.json")))
    assert len(json.dumps(full)) > 20000
    line, text = bench.contract_line(full, "gpurun_out/bench_full.json (+ stderr)")
    assert len(text.encode()) < 4096 and "\n" not in text
    assert json.loads(text) == json.loads(json.dumps(line))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "e2e_value", "e2e_frac_of_kernel", "kernel_value"):
        assert key in line, key
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["e2e_value"] == pytest.approx(full["end_to_end"]["C2 3xCNN L=8 list_str"]["value"], rel=1e-5)
    assert "workload" in line["config"] and "model" not in line["config"] and "path" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 157.3 and 0 < roof["frac"] <= 1.0 and roof["kernel"] == "k_score_cnn_mfma"
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3) and isinstance(roof["traffic"], int)
    # nothing nested, no prose beyond the names: a parser that keeps only scalars loses nothing
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert all(not isinstance(v, (dict, list)) for v in line["cpu_baseline"].values())
    for k in ("c1_kernel_ms", "c1_frac_issued", "c2_1e4_frac_issued", "c3_kernel_ms", "c3_frac_issued", "c4_frac_issued", "c5_frac_issued",
              "mlp_h200_frac_issued", "cnn_h200_frac_issued", "ge_m1_frac_issued", "k4_c100", "k4_c1000", "k4_c20000",
              "settled_kernel_ms", "settled_frac_issued", "e2e_c2_frac_of_kernel", "e2e_c3_frac_of_kernel", "e2e_c4_frac_of_kernel",
              "e2e_c5_seq_per_s", "train_l8_ms", "train_l237_ms", "train_l237_frac_of_peak", "small_call_n20_us", "dynappo_n10_us", "cmaes_p40_us"):
        assert isinstance(roof.get(k), float) and roof[k] > 0, k
    assert roof["c1_kernel_ms"] == pytest.approx(full["configs"]["C1 cnn L=8 A=4 M=1 N=1e4"]["kernel_ms"], rel=1e-3)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["vectorised_value"] > 0 and len(cb["sample"]) <= 200
    # a record that grew (a future block with many scalars) still gives a line under the limit with every contract key
    fat = json.loads(json.dumps(full))
    for i in range(400):
        fat["configs"][f"survey extra {i}"] = {"kernel_ms": 1.0, "frac_issued": 0.5}
    bench.CONFIG_TAGS.update({f"x{i}": f"survey extra {i}" for i in range(400)})
    try:
        fline, ftext = bench.contract_line(fat, None)
    finally:
        for i in range(400):
            bench.CONFIG_TAGS.pop(f"x{i}")
    assert len(ftext.encode()) <= 4096 and fline["roofline"]["frac"] == roof["frac"] and "cpu_baseline" in fline
    # a headline-only record (--no-extras, N > 1 ranks) has no verbose blocks at all
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "roofline")}
    bline, btext = bench.contract_line(bare, None)
    assert "e2e_value" not in bline and "cpu_baseline" not in bline and json.loads(btext)["roofline"]["frac"] == roof["frac"]
    # the child that times the train_swizzle forms (--prepared): whatever goes wrong in it is a field of the record, never an exception
    from tools import bench_blocks

    got = bench_blocks.prepared_block(timeout_s=120.0)
    assert isinstance(got, dict) and ("error" in got or "skipped" in got or any("ms_per_fit" in v for v in got.values() if isinstance(v, dict)))
    json.dumps(got)


def test_bench_prints_the_contract_line_last_on_stdout(tmp_path):
    """`bench.py --cpu-selftest` (gloo, injected scorer: the launch path without a GPU) prints exactly one stdout line, valid JSON,
    under the limit -- the same print path the GPU run ends with."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-selftest"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 4096
    assert json.loads(lines[0])["selftest"]["member"]["ok"] is True


def test_nam_device_mirror_follows_the_cache_without_walking_it():
    """`NoisyAbstractModel._sync_device_cache` (round 5): the keys the model itself adds are noted as they are added, so bringing the
    device mirror up to date no longer skips the first n keys of the dict (O(cache) per call).  The mirror must still equal
    `list(cache)` whatever happens to the dict: the model's own inserts (duplicates inside a batch, keys it already has), a key put in
    by somebody else, deletions, a replaced dict, a pickle round trip."""
    from flexs_amd.baselines.models import noisy_abstract_model as nm

    class FakeCache:                       # what _new_device_cache hands out: records the rows appended (no GPU here)
        def __init__(self, L):
            self.L, self.rows = L, []

        def append(self, rows):
            self.rows.extend(bytes(r).rstrip(b"\x00").decode() for r in rows)

    class Land(flexs_amd.Landscape):
        def __init__(self):
            super().__init__("land")

        def _fitness_function(self, seqs):
            return np.array([len(s) / 10.0 for s in seqs])

    class M(nm.NoisyAbstractModel):
        def _new_device_cache(self, row_bytes):
            return FakeCache(row_bytes)

        def _rows(self, seqs, L):
            return [s.encode().ljust(L, b"\x00") for s in seqs]

    m = M(Land(), 0.9)

    def synced():
        m._sync_device_cache(1)
        assert m._dev_keys == [str(k) for k in m.cache] == m._dev_cache.rows, (m._dev_keys, list(m.cache))

    m.train(["AAA", "CCC", "AAA", "GGG"], np.array([1.0, 2.0, 3.0, 4.0]))
    assert m._pending == ["AAA", "CCC", "GGG"]
    synced()
    m.train(np.array(["TTT", "CCC", "ACG"]), np.array([5.0, 6.0, 7.0]))      # np.str_ keys, one of them known
    assert m._pending == ["TTT", "ACG"]
    synced()
    # the reference's train is `cache.update(zip(sequences, labels))`: a one-shot iterable (generator, map) is legal input and must not be
    # used up by the note pass (round-5 advisor); an instance without `_pending` (older pickle, subclass skipping __init__) re-walks the dict
    m.train((s_ for s_ in ["GTT", "GTA"]), np.array([8.0, 9.0]))
    assert m.cache["GTT"] == 8.0 and m.cache["GTA"] == 9.0 and m._pending == ["GTT", "GTA"]
    synced()
    del m._pending
    m.train(map(str, ["CGT"]), np.array([1.5]))
    assert m.cache["CGT"] == 1.5
    synced()
    m._note_new_keys(["GGA", "GGA", "AAA"]); m.cache.update(zip(["GGA", "GGA", "AAA"], [1.0, 2.0, 3.0]))
    assert m._pending == ["GGA"]
    m.cache["CAT"] = 0.5                   # somebody else writes to the public dict: the note is out of step, the walk takes over
    synced()
    m._note_new_keys(["TAG"]); m.cache.update({"TAG": 0.1})
    del m.cache["CCC"]                     # a removal anywhere: rebuild
    synced()
    m.cache = dict(m.cache); m.cache["NEW"] = 1.0      # a plain dict instead of the counting one
    m._note_new_keys(["XYZ"]); m.cache["XYZ"] = 2.0
    synced()
    m._note_new_keys(["QQQ"]); m.cache["QQQ"] = 3.0
    state = m.__getstate__()               # what copy / pickle carry: no device mirror, no pending note
    assert state["_pending"] == [] and state["_dev_keys"] == [] and state["_dev_cache"] is None and "QQQ" in state["cache"]


def test_staged_packing_fills_a_tile_pitched_area_and_publishes_every_lane():
    """`_strpack.pack_staged` (the host half of a launched-first call, include/flexs_amd.h fx_score_begin_staged): the rows of tile t
    land at t * pitch, every lane's word ends at base + stages whatever happens, the statuses are `pack`'s."""
    from flexs_amd import _native, synth
    sp = _native._strpack
    if sp is None or not hasattr(sp, "pack_staged"):
        pytest.skip("_strpack was not built")
    for n, L, Q, lanes in ((100_000, 14, 7, 8), (1000, 8, 3, 4), (33, 5, 2, 1), (100_001, 8, 19, 8), (5000, 90, 4, 16), (70, 14, 3, 2),
                           (40, 8, 9, 3)):                       # (more stages than tiles: the empty ones are published too)
        ref = np.asarray(synth.random_sequence_bytes(n, L, "ACGT", 3)).reshape(n, L)
        seqs = synth.bytes_to_strings(ref)
        pitch, TG = (16 * L + 127) // 128 * 128, (n + 15) // 16
        dst = np.full(TG * pitch, 0xEE, np.uint8)
        words = np.zeros(16, np.uint32)
        assert sp.pack_staged(seqs, L, dst.ctypes.data, Q, pitch, lanes, words.ctypes.data, 4096) == 0
        tiles = dst.reshape(TG, pitch)
        assert (tiles[:, :16 * L].reshape(TG * 16, L)[:n] == ref).all()
        assert (tiles[:, 16 * L:] == 0xEE).all()                 # the padding of a tile is never written
        assert (words[:lanes] == 4096 + Q).all() and (words[lanes:] == 0).all()
        for pos, bad, status in ((n // 2, "A" * (L + 1), 1), (n - 1, "A" * (L - 1) + "\u0394", 2), (0, 7, 3)):
            broken = list(seqs)
            broken[pos] = bad
            words[:] = 0
            assert sp.pack_staged(broken, L, dst.ctypes.data, Q, pitch, lanes, words.ctypes.data, 8192) == status
            assert (words[:lanes] == 8192 + Q).all()             # the kernels are never left waiting
    assert sp.lanes_for(100) == 1 and 1 <= sp.lanes_for(1 << 24) <= 16
    with pytest.raises(ValueError):
        sp.pack_staged(["ACGT"], 4, 1, 2, 16, 1, 1, 0)           # pitch < 16 L
