"""GPU tests of the drop-in Python API and the C ABI's behaviour around the kernels: cost bookkeeping, error mapping, population
step, one-rank distributed classes, big string batches, misuse, copies / pickles of live models, engine counters."""
import json
import os

import numpy as np
import pytest

import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
from oracle import c_oracle, ref_np

from gpu_common import ATOL, ERROR_STATS, RTOL, ab_option, assert_scores, close, eng, make_native, rand_seqs  # noqa: F401  (eng: the session fixture)

pytestmark = pytest.mark.gpu


def test_python_api_drop_in(eng):
    """flexs.Model surface: dtypes, cost accounting (ensemble.py:55-57 via landscape.py:44), names."""
    L, alpha = 14, "UGCA"
    b, seqs = rand_seqs(333, L, alpha, seed=2)
    members = [bm.CNN(L, 32, 100, alpha, seed=0), bm.MLP(L, 100, alpha, seed=1), bm.GlobalEpistasisModel(L, 100, alpha, seed=2)]
    for m, kind in zip(members, ("cnn", "mlp", "ge")):
        out = m.get_fitness(seqs)
        assert out.dtype == np.float32 and out.shape == (333,) and m.cost == 333
        assert_scores(out, ref_np.keras_fitness(seqs, alpha, kind, m.model.get_weights(), exact=True), kind)
        assert np.array_equal(m.get_fitness(np.array(seqs)), out)            # ndarray input
        assert np.array_equal(m.get_fitness(tuple(seqs[:5])), out[:5])
        assert m.get_fitness([]).shape == (0,)
    ens = flexs_amd.Ensemble(members)
    for m in members:
        m.cost = 0
    out = ens.get_fitness(seqs)
    assert ens.cost == 333 and all(m.cost == 333 for m in members)
    stack = np.stack([m.get_fitness(seqs) for m in members], axis=1)
    assert np.array_equal(out, np.mean(stack, axis=1))
    ident = flexs_amd.Ensemble(members, combine_with=lambda x: x).get_fitness(seqs)   # BO's usage (bo.py:55-56)
    assert np.array_equal(ident, stack)
    ada = bm.AdaptiveEnsemble(members)
    assert np.array_equal(ada.get_fitness(seqs), np.sum(ada.weights * stack, axis=1))
    # weights reload (once per explorer round): set_weights must reach the device
    new_w = ref_np.synth_weights(ref_np.mlp_shapes(L, 4, 100), 99)
    members[1].model.set_weights(new_w)
    assert_scores(members[1].get_fitness(seqs), ref_np.keras_fitness(seqs, alpha, "mlp", new_w, exact=True), "reloaded")
    # the reference's own smoke scenario (tests/test_models.py:55-77)
    bm.CNN(seq_len=3, num_filters=1, hidden_size=1, kernel_size=2, alphabet=s_utils.DNAA).get_fitness(["ATC"])
    bm.GlobalEpistasisModel(seq_len=3, hidden_size=1, alphabet=s_utils.DNAA).get_fitness(["ATC"])
    bm.MLP(seq_len=3, hidden_size=1, alphabet=s_utils.DNAA).get_fitness(["ATC"])


def test_errors(eng):
    cnn = bm.CNN(8, 32, 100, "TGCA", seed=0)
    with pytest.raises(ValueError):
        cnn.get_fitness(["ATGCATGX"])                      # str.index ValueError (sequence_utils.py:46)
    assert cnn.get_fitness(["ATGCATGC"]).shape == (1,)     # engine still usable afterwards
    with pytest.raises(ValueError):
        cnn.get_fitness(["ATGC"])                          # wrong length
    with pytest.raises(ValueError):
        cnn.get_fitness(["ATGCATGC", "ATG"])               # ragged
    with pytest.raises(ValueError):
        s_utils.string_to_one_hot("ATXG", s_utils.DNAA)
    lowercase = bm.MLP(4, 8, "TGCA", seed=0)
    with pytest.raises(ValueError):
        lowercase.get_fitness(["atgc"])
    # nan_to_num (keras_model.py:77)
    w = lowercase.model.get_weights()
    w[-1][:] = np.nan
    lowercase.model.set_weights(w)
    assert lowercase.get_fitness(["ATGC"]).tolist() == [0.0]
    w[-1][:] = np.inf
    lowercase.model.set_weights(w)
    assert lowercase.get_fitness(["ATGC"]).tolist() == [float(np.finfo(np.float32).max)]


# ------------------------------------------------------------------ population step (section 8f-2)
@pytest.mark.parametrize("which", ["ensemble", "single", "host-stacked"])
def test_population_evaluator_equals_one_by_one_loop(eng, which):
    """cmaes.py:61-67 + 83-93 for a whole population at once == the reference's loop of
    `get_fitness([seq]).item()` calls: strings, values (bit for bit), cost on the model and its members."""
    from flexs_amd.utils.population import PopulationEvaluator

    L, alpha, P = 8, "TGCA", 37
    rng = np.random.default_rng(5)

    def build():
        members = [bm.CNN(L, 32, 100, alpha, seed=s) for s in range(3)]
        if which == "single":
            return members[0], [members[0]]
        if which == "host-stacked":                      # custom reduction: not fused, answered one by one
            return flexs_amd.Ensemble(members, combine_with=lambda x: np.median(x, axis=1)), members
        return flexs_amd.Ensemble(members), members

    model, members = build()
    twin, twin_members = build()
    x = rng.standard_normal((P, L * len(alpha)))
    x[5] = x[2]                                          # duplicate solutions inside one population
    x[9, :4] = 0.0                                       # a tie: first maximum wins
    want_seqs = [ref_np.one_hot_to_string(r.reshape(L, len(alpha)), alpha) for r in x]
    known_a = {want_seqs[0]: 123.0, want_seqs[7]: -1.5}
    known_b = {want_seqs[0]: 999.0, want_seqs[11]: 0.25}
    want_vals = []
    for s in want_seqs:                                  # objective_function, cmaes.py:83-93
        if s in known_a:
            want_vals.append(known_a[s])
        elif s in known_b:
            want_vals.append(known_b[s])
        else:
            want_vals.append(twin.get_fitness([s]).item())
    ev = PopulationEvaluator(model, alpha, L)
    assert ev.decode(x) == want_seqs
    seqs, vals = ev.evaluate(x, known=(known_a, known_b))
    assert seqs == want_seqs and vals.dtype == np.float64 and vals.tolist() == want_vals
    # ... and the values themselves against the ORACLE (the loop above compares the HIP path with itself)
    fresh = [i for i, s_ in enumerate(want_seqs) if s_ not in known_a and s_ not in known_b]
    stack = np.stack([ref_np.keras_fitness([want_seqs[i] for i in fresh], alpha, "cnn", m.model.get_weights(), exact=True)
                      for m in members], axis=1)
    want_oracle = stack[:, 0] if which == "single" else (np.median(stack, axis=1) if which == "host-stacked" else stack.mean(axis=1))
    assert_scores(vals[fresh].astype(np.float32), want_oracle, f"population values ({which})")
    assert model.cost == twin.cost == P - 3
    if which != "single":
        assert [m.cost for m in members] == [m.cost for m in twin_members] == [P - 3] * 3
    assert ev.evaluate(np.zeros((0, L * 4)))[0] == []
    # no `known` dicts (DyNA-PPO's environment step): argmax + scoring + strings in one C call (strpack.population_step) == the step in
    # pieces (host argmax, Engine.score, per-row str) == the device argmax form, values and cost
    from flexs_amd.utils import population
    c0 = model.cost
    seqs1, vals1 = ev.evaluate(x)
    assert seqs1 == want_seqs and model.cost == c0 + P
    helper = _native._strpack.population_step if which != "host-stacked" and hasattr(_native._strpack, "population_step") else None
    try:
        if helper is not None:
            del _native._strpack.population_step             # (the step in pieces)
        seqs2, vals2 = ev.evaluate(x)
        population.HOST_DECODE = False                       # (argmax on the device: fx_decode_score)
        seqs3, vals3 = ev.evaluate(x)
    finally:
        population.HOST_DECODE = True
        if helper is not None:
            _native._strpack.population_step = helper
    assert seqs2 == want_seqs and seqs3 == want_seqs
    assert vals1.tolist() == vals2.tolist() == vals3.tolist()
    assert vals1.tolist() == [twin.get_fitness([s_]).item() for s_ in want_seqs]
    if which != "host-stacked":
        with pytest.raises(ValueError):
            PopulationEvaluator(model, "UGCA", L)


def test_terminal_rewards_equal_environment_loop(eng):
    """environments/dyna_ppo.py:106-114 + 144-163 for a whole environment batch: same sequences, fitnesses and
    density-penalised rewards as the per-sequence Python loops (density counted after the batch is recorded)."""
    from flexs_amd.utils.edit_distance import SeenSequences
    from flexs_amd.utils.population import PopulationEvaluator, terminal_rewards

    L, alpha, B, lam = 14, "UGCA", 24, 0.1
    rng = np.random.default_rng(6)
    model = flexs_amd.Ensemble([bm.MLP(L, 100, alpha, seed=s) for s in range(2)])
    twin = flexs_amd.Ensemble([bm.MLP(L, 100, alpha, seed=s) for s in range(2)])
    seen, all_seqs = SeenSequences(L), {}
    ev = PopulationEvaluator(model, alpha, L)
    base = rng.integers(0, 4, L)
    for episode in range(3):
        states = np.zeros((B, L, len(alpha) + 1))
        for b in range(B):
            codes = base.copy()
            m = rng.random(L) < 0.1
            codes[m] = rng.integers(0, 4, m.sum())
            states[b, np.arange(L), codes] = 1
        states[1] = states[0]                                         # duplicates inside one batch
        seqs, fit, rew = terminal_rewards(ev, seen, states, lam)
        want_seqs = [ref_np.one_hot_to_string(st[:, :-1], alpha) for st in states]
        want_fit = twin.get_fitness(want_seqs)
        all_seqs.update(zip(want_seqs, want_fit.astype(np.float64)))
        want_rew = []
        for s_, f in zip(want_seqs, want_fit.astype(np.float64)):
            dens = 0
            for k in all_seqs:
                dist = c_oracle.levenshtein(k, s_)
                if dist != 0 and dist <= 2:
                    dens += all_seqs[k] / dist
            want_rew.append(f - lam * dens)
        assert seqs == want_seqs and fit.tolist() == want_fit.astype(np.float64).tolist()
        assert rew.tolist() == want_rew
        assert model.cost == twin.cost and len(seen) == len(all_seqs)


def test_distributed_classes_on_one_rank_rccl(eng):
    """flexs_amd.distributed over a real one-rank RCCL group (backend "nccl"): the device all-gather, weight
    broadcast and the default on-engine scorers -- what the gloo tests replace by stubs -- give the single-GPU
    Ensemble / cache answers."""
    import socket

    import torch
    import torch.distributed as dist

    from flexs_amd import distributed as fd

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        L, alpha = 14, "UGCA"
        members = [bm.CNN(L, 32, 100, alpha, seed=0), bm.CNN(L, 32, 100, alpha, seed=1), bm.CNN(L, 32, 100, alpha, seed=2)]
        b, seqs = rand_seqs(257, L, alpha, seed=4)
        want = flexs_amd.Ensemble(members).get_fitness(seqs)
        stack = np.stack([m.get_fitness(seqs) for m in members], axis=1)
        for mode in ("member", "sequence"):
            for force in (True, False):              # the real RCCL all-gather on device buffers / the one-rank alias
                ens = fd.DistributedEnsemble(members, mode=mode)
                ens.force_collective = force
                assert np.array_equal(ens.get_fitness(seqs), want)
                mat = fd.DistributedEnsemble(members, mode=mode, combine_with=lambda x: x)
                mat.force_collective = force
                assert np.array_equal(mat.get_fitness(seqs), stack)
                ens.broadcast_weights(src=0)
                assert np.array_equal(ens.get_fitness(seqs), want)
                # the two halves on a batch already resident in HBM, both buffer slots in flight (what bench.py does)
                with torch.cuda.stream(ens.stream):
                    d_seq = torch.from_numpy(b).cuda()
                ens.launch(d_seq, slot=0, want="mean")
                ens.launch(d_seq, slot=1, want="matrix")
                got_mean, got_mat = ens.finish(0), ens.finish(1)
                ens.stream.synchronize()
                assert np.array_equal(got_mean.cpu().numpy(), want) and np.array_equal(got_mat.cpu().numpy(), stack)
                with pytest.raises(ValueError):
                    ens.get_fitness(seqs[:5] + ["Z" * L])
                assert np.array_equal(ens.get_fitness(seqs), want)
            assert ens.cost == 3 * 257 + 6 and all(m.cost > 0 for m in members)
        sc = fd.ShardedCache(L)
        sc.append(b[:200])
        d, a = sc.min_dist(b[150:])
        d_want, a_want = c_oracle.min_dist(b[150:], b[:200], 0)
        assert np.array_equal(d, d_want) and np.array_equal(a, a_want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,L,alpha,M,n", [("cnn", 8, "TGCA", 8, 1000), ("ge", 90, s_utils.AAS, 8, 333), ("mlp", 14, "UGCA", 17, 65)])
def test_distributed_ensemble_without_a_process_group(eng, kind, L, alpha, M, n):
    """No torch.distributed at all (world = 1): DistributedEnsemble is the device-resident path of a plain Ensemble --
    planes in HBM, K3 on the planes, only the result copied back -- and must give Ensemble's bits."""
    from flexs_amd import distributed as fd

    mk = {"cnn": lambda s: bm.CNN(L, 32, 100, alpha, seed=s), "ge": lambda s: bm.GlobalEpistasisModel(L, 100, alpha, seed=s),
          "mlp": lambda s: bm.MLP(L, 100, alpha, seed=s)}[kind]
    members = [mk(s) for s in range(M)]
    b, seqs = rand_seqs(n, L, alpha, seed=11)
    want = flexs_amd.Ensemble(members).get_fitness(seqs)
    stack = flexs_amd.Ensemble(members, combine_with=lambda x: x).get_fitness(seqs)
    for mode in ("member", "sequence"):
        assert np.array_equal(fd.DistributedEnsemble(members, mode=mode).get_fitness(seqs), want)
        assert np.array_equal(fd.DistributedEnsemble(members, mode=mode, combine_with=lambda x: x).get_fitness(seqs), stack)
    assert fd.DistributedEnsemble(members).get_fitness([]).shape == (0,)


def test_big_string_batches_are_scored_in_overlapping_pieces(eng):
    """list[str] batches of >= 16384 sequences take the launched-first or the chunked host call (fx_score_begin / _submit / _finish):
    same scores, cost accounting and exceptions as the one-piece call."""
    L, alpha, N = 8, "TGCA", 70_001
    members = [bm.CNN(L, 32, 100, alpha, seed=s) for s in range(3)]
    ens = flexs_amd.Ensemble(members)
    b, seqs = rand_seqs(N, L, alpha, seed=21)
    assert _native.wants_chunked(seqs, L) and not _native.wants_chunked(seqs[:100], L)
    want_nm, want_mean = eng.score([m.native() for m in members], b, members[0]._lut, want_matrix=True, want_mean=True)
    assert np.array_equal(ens.get_fitness(seqs), want_mean)                      # chunked, fused mean
    assert np.array_equal(ens.get_fitness(tuple(seqs)), want_mean)
    assert np.array_equal(members[1].get_fitness(seqs), want_nm[:, 1])           # chunked, single model
    assert np.array_equal(flexs_amd.Ensemble(members, combine_with=lambda x: x).get_fitness(seqs), want_nm)
    assert ens.cost == 2 * N and members[0].cost == 3 * N and members[1].cost == 4 * N
    for chunks in (1, 3, 7):
        nm, mean = eng.score_strings([m.native() for m in members], seqs, L, members[0]._lut, True, True, chunks=chunks)
        assert np.array_equal(nm, want_nm) and np.array_equal(mean, want_mean)
    for pos, bad, exc in ((N - 5, "TGCAZGCA", ValueError), (N - 5, "TGCA", ValueError), (60_000, 7, TypeError),
                          (3, "TGCATΔCA", ValueError)):
        broken = list(seqs)
        broken[pos] = bad
        with pytest.raises(exc):
            ens.get_fitness(broken)
        assert np.array_equal(ens.get_fitness(seqs[:40_000]), want_mean[:40_000])   # the engine is usable afterwards


@pytest.mark.parametrize("kind,L,alpha,M,n", [("cnn", 8, "TGCA", 3, 100_001), ("cnn", 8, "TGCA", 1, 5), ("mlp", 14, "UGCA", 8, 1003),
                                              ("ge", 90, s_utils.AAS, 16, 257), ("cnn", 237, s_utils.AAS, 2, 40),
                                              ("cnn", 9, "TGCA", 3, 77), ("mlp", 14, "UGCA", 17, 300)])
def test_mean_only_path_uses_planes_and_matches_matrix_path(eng, kind, L, alpha, M, n):
    """Asking for the mean only lets the engine keep the scores as member-major planes (contiguous stores);
    the mean must equal np.mean over the (N, M) matrix of the other path bit for bit, for every kernel family
    (MFMA, pair / segmented, shape-agnostic) and through the explicit two-call form."""
    import torch

    F, K = (32, 5) if kind == "cnn" else (0, 0)
    if L == 9:
        F, K = 8, 4                                             # shape-agnostic kernels
    natives, _ = zip(*[make_native(eng, kind, L, len(alpha), 100 if L != 9 else 20, F, K, seed=300 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, _ = rand_seqs(n, L, alpha, seed=n)
    nm, mean_a = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)      # row-major intermediate
    _, mean_b = eng.score(list(natives), b, lut, want_matrix=False, want_mean=True)      # planes (M <= 16)
    assert np.array_equal(mean_a, np.mean(nm, axis=1)) and np.array_equal(mean_b, mean_a)
    d_in = torch.from_numpy(b).cuda()
    d_mean = torch.full((n,), float("nan"), device="cuda")
    eng.score_dev(list(natives), d_in.data_ptr(), n, L, lut, None, d_mean.data_ptr())
    eng.sync()
    assert np.array_equal(d_mean.cpu().numpy(), mean_a)
    if M <= 16:
        stride = (n + 63) // 64 * 64
        planes = torch.full((M, stride), float("nan"), device="cuda")
        d_mean.fill_(float("nan"))
        eng.score_planes_dev(list(natives), d_in.data_ptr(), n, L, lut, planes.data_ptr(), stride)
        eng.ensemble_mean_planes_dev(planes.data_ptr(), n, M, stride, d_mean.data_ptr())
        eng.sync()
        assert np.array_equal(planes[:, :n].t().cpu().numpy(), nm) and np.array_equal(d_mean.cpu().numpy(), mean_a)
        off = torch.full((n + 1,), float("nan"), device="cuda")       # a destination that is not 16-byte aligned
        eng.ensemble_mean_planes_dev(planes.data_ptr(), n, M, stride, off.data_ptr() + 4)
        eng.sync()
        assert np.array_equal(off[1:].cpu().numpy(), mean_a)


def test_c_abi_misuse_is_reported_not_fatal(eng):
    """Status codes of the C ABI on misuse: every call returns an fx_status (mapped to ValueError / FxError by the
    Python layer), nothing aborts, and the engine keeps working afterwards."""
    import ctypes as C

    lib, h = eng._lib, eng.handle
    lut = _native.make_lut("TGCA")
    nm, w = make_native(eng, "cnn", 8, 4, 100, 32, 5, seed=1)
    b, _ = rand_seqs(32, 8, "TGCA", seed=1)
    good, _ = eng.score([nm], b, lut)
    # weights never set
    empty = _native.NativeModel(eng, _native.FX_CNN, 8, 4, 32, 100, 5)
    with pytest.raises(_native.FxError) as err:
        eng.score([empty], b, lut)
    assert err.value.code == _native.FX_ESTATE
    # wrong sequence length for the model -> ValueError (Keras shape error)
    with pytest.raises(ValueError):
        eng.score([nm], b[:, :7].copy(), lut)
    # LUT that maps a byte beyond the alphabet
    bad_lut = lut.copy(); bad_lut[ord("Z")] = 9
    with pytest.raises(_native.FxError) as err:
        eng.score([nm], b, bad_lut)
    assert err.value.code == _native.FX_EINVAL
    # members with different alphabets / a valid-conv that cannot exist / wrong weight count
    with pytest.raises(ValueError):
        eng.score([nm, make_native(eng, "cnn", 8, 20, 100, 32, 5, seed=2)[0]], b, lut)
    with pytest.raises((ValueError, _native.FxError)):
        _native.NativeModel(eng, _native.FX_CNN, 3, 4, 32, 100, 5)                     # L < kernel_size
    with pytest.raises((ValueError, _native.FxError)):
        nm.set_weights(w[:-1])
    # raw calls: null buffers, negative sizes, unknown option, protocol errors
    arr = (C.c_void_p * 1)(nm.handle)
    assert lib.fx_score(h, arr, 1, None, 4, 8, lut.ctypes.data_as(_native._u8p), None, None) == _native.FX_EINVAL
    assert lib.fx_score(h, arr, 1, None, -1, 8, lut.ctypes.data_as(_native._u8p), None, None) == _native.FX_EINVAL
    assert lib.fx_score(h, arr, 0, None, 4, 8, lut.ctypes.data_as(_native._u8p), None, None) == _native.FX_EINVAL
    assert lib.fx_engine_set_option(h, b"no_such_option", 1) != _native.FX_OK
    assert lib.fx_score_submit(h, 0, 16) == _native.FX_ESTATE and lib.fx_score_finish(h, None, None) == _native.FX_ESTATE
    assert b"fx_score_finish" in lib.fx_last_error(h)
    assert lib.fx_min_dist(h, 0, None, 4, None, 4, 800, None, None) != _native.FX_OK      # null buffers
    # (rows beyond 768 symbols are served since round 3 -- the strip form of the recurrence, csrc/mindist.hip)
    d800, a800 = eng.min_dist(np.zeros((2, 800), np.uint8) + 65, np.zeros((3, 800), np.uint8) + 65)
    assert (d800 == 0).all() and (a800 == 0).all()
    with pytest.raises((ValueError, _native.FxError)):
        _native.NativeTable(eng, np.zeros((4, 5)), "", lut=np.full(256, 7, np.uint8)).additive_sum(np.zeros((2, 4), np.uint8))
    assert lib.fx_status_name(_native.FX_EBADCHAR) == b"FX_EBADCHAR" and lib.fx_version() >= 100
    # ... and the engine still scores
    again, _ = eng.score([nm], b, lut)
    assert np.array_equal(again, good)
    # non-finite weights inside the network: NaN / inf end as nan_to_num says (keras_model.py:77)
    w2 = [x.copy() for x in w]
    w2[2][0, 0, 0] = np.nan                                                            # a conv2 weight
    nm.set_weights(w2)
    out, _ = eng.score([nm], b, lut)
    assert np.isfinite(out).all()


def test_deepcopy_and_pickle_of_live_models(eng):
    """Models that already own device handles can be deep-copied and pickled (an explorer wrapper might): the copy
    re-creates its own handles lazily and scores identically; NoisyAbstractModel rebuilds its device key store."""
    import copy
    import pickle

    L, alpha = 14, "UGCA"
    _, seqs = rand_seqs(300, L, alpha, seed=77)
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alpha, seed=0), bm.MLP(L, 100, alpha, seed=1)])
    want = ens.get_fitness(seqs)                                          # handles now exist
    for clone in (copy.deepcopy(ens), pickle.loads(pickle.dumps(ens))):
        assert clone.models[0]._native_model is None
        assert np.array_equal(clone.get_fitness(seqs), want) and clone.cost == 600

    class Table(flexs_amd.Landscape):
        def __init__(self):
            super().__init__("table")

        def _fitness_function(self, s):
            return np.array([(sum(map(ord, str(x))) % 97) / 97.0 for x in s])

    nam = bm.NoisyAbstractModel(Table(), 0.8)
    nam.train(seqs[:100], np.linspace(0, 1, 100))
    np.random.seed(1)
    nam.get_fitness(seqs[100:150])
    twin = copy.deepcopy(nam)
    np.random.seed(2); a = nam.get_fitness(seqs[150:220])
    np.random.seed(2); b = twin.get_fitness(seqs[150:220])
    assert np.array_equal(a, b) and list(nam.cache) == list(twin.cache)


def test_engine_counters(eng):
    """fx_engine_counters: the engine's own account of what went through it (SURVEY.md section 5 aux: counters) -- host
    calls, zero-copy vs copy path bytes, forwards, distance evaluations, training steps."""
    from flexs_amd import training

    eng.counters(reset=True)
    L, alpha = 8, "TGCA"
    members = [bm.CNN(L, 32, 100, alpha, seed=s) for s in range(3)]
    ens = flexs_amd.Ensemble(members)
    b, seqs = rand_seqs(20, L, alpha, seed=1)
    ens.get_fitness(seqs)                                         # explorer-size call: zero-copy
    c = eng.counters()
    assert c["host_calls"] == 1 and c["zero_copy_calls"] == 1 and c["sequences"] == 20 and c["forwards"] == 60 and c["bytes_h2d"] == 0
    eng.set_option("zero_copy_mode", 0)                           # force the copy path for a big batch
    try:
        b2, _ = rand_seqs(50_000, L, alpha, seed=2)
        eng.score([m.native() for m in members], b2, members[0]._lut, want_matrix=True)
    finally:
        eng.set_option("zero_copy_mode", -1)
    c = eng.counters()
    assert c["host_calls"] == 2 and c["zero_copy_calls"] == 1 and c["bytes_h2d"] == 50_000 * L and c["bytes_d2h"] == 4 * 3 * 50_000
    assert c["sequences"] == 50_020 and c["forwards"] == 3 * 50_020
    eng.min_dist(b2[:7], b2[:1000])
    assert eng.counters()["pair_evals"] == 7000
    y = np.random.default_rng(0).random(20)
    if training._train_mode(__import__("torch").device("cuda")) == "native":
        ens.train(seqs, y)
        assert eng.counters()["train_steps"] == 3 * 20            # 20 epochs x 1 mini-batch x 3 members
    assert eng.counters(reset=True)["host_calls"] == 2 and eng.counters()["host_calls"] == 0
