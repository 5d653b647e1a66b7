"""World-size-2 `gloo` tests of the multi-GPU sharding / gather logic (runs on CPU).
The scorer is injected (a deterministic table function), so what is tested is
exactly what differs from the single-GPU path: shard ranges, member assignment,
the padded all-gather, re-assembly order and cost accounting."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import flexs_amd
from flexs_amd import distributed as fd


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 100_000, 100_003):
        for world in (1, 2, 3, 8):
            spans = [fd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert fd.member_assignment(8, 1, 8) == [1] and fd.member_assignment(3, 1, 2) == [2]
    assert fd.member_assignment(3, 5, 8) == [] and fd.member_assignment(8, 1, 4) == [2, 3]
    assert sorted(sum((fd.member_assignment(11, r, 4) for r in range(4)), [])) == list(range(11))


class Stub(flexs_amd.Model):
    """A member whose score is a pure function of (member id, sequence bytes)."""

    def __init__(self, mid):
        super().__init__(f"stub{mid}")
        self.mid = mid

    def train(self, *a):
        pass

    def _fitness_function(self, sequences):
        raise AssertionError("members are scored through score_fn in this test")


def table_score(member_idx, seq_bytes):
    s = seq_bytes.astype(np.float64)
    w = np.arange(1, s.shape[1] + 1)
    return np.stack([np.sin((s * w).sum(axis=1) * (m + 1) * 1e-3) for m in member_idx], axis=1).astype(np.float32) \
        if member_idx else np.zeros((s.shape[0], 0), np.float32)


def _collect(q, procs):
    results = []
    for _ in procs:
        r = q.get(timeout=120)
        assert len(r) > 2, f"rank {r[0]} failed: {r[1]}"
        results.append(r)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    return sorted(results, key=lambda t: t[0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, M, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        seqs = ["".join("TGCA"[i] for i in row) for row in rng.integers(0, 4, (n, 8))]
        members = [Stub(m) for m in range(M)]
        called = []

        def score_fn(idx, b):
            called.append((list(idx), b.shape[0]))
            return table_score(idx, b)

        ens = fd.DistributedEnsemble(members, mode=mode, score_fn=score_fn)
        out = ens.get_fitness(seqs)
        mat = fd.DistributedEnsemble(members, mode=mode, score_fn=score_fn, combine_with=lambda x: x).get_fitness(seqs)
        q.put((rank, out, mat, called, ens.cost, [m.cost for m in members]))
    except BaseException as exc:      # surface worker failures at once instead of after the queue timeout
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,M,n", [("member", 3, 101), ("member", 8, 64), ("sequence", 3, 101), ("sequence", 2, 1),
                                      ("member", 1, 5), ("member", 17, 130), ("sequence", 17, 67)])
def test_world2_gloo(mode, M, n):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, M, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs)
    rng = np.random.default_rng(0)
    seq_bytes = np.frombuffer(b"TGCA", np.uint8)[rng.integers(0, 4, (n, 8))]
    want_mat = table_score(list(range(M)), seq_bytes)
    for rank, out, mat, called, cost, mcosts in results:
        assert np.array_equal(mat, want_mat)                       # every rank holds the full stacked matrix
        assert np.array_equal(out, np.mean(want_mat, axis=1))
        assert cost == 2 * n // 2 and mcosts == [2 * n] * M        # two get_fitness calls on members, one per ensemble
        if mode == "member":
            want_idx, want_rows = fd.member_assignment(M, rank, world), n
        else:
            lo, hi = fd.shard_range(n, rank, world)
            want_idx, want_rows = list(range(M)), hi - lo
        if want_idx and want_rows:                                 # a rank with nothing to score launches nothing
            assert called[0] == (want_idx, want_rows) and len(called) == 2
        else:
            assert called == []


def _count_collectives():
    """Wrap every torch.distributed collective the module could issue; returns the list the calls are recorded in."""
    calls = []
    for name in ("all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "barrier", "reduce", "gather", "all_to_all"):
        orig = getattr(dist, name)

        def counted(*a, _orig=orig, _name=name, **kw):
            calls.append(_name)
            return _orig(*a, **kw)

        setattr(dist, name, counted)
    return calls


def _one_collective_worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        n = 101
        seqs = ["".join("TGCA"[i] for i in row) for row in rng.integers(0, 4, (n, 8))]
        members = [Stub(m) for m in range(3)]
        bad_on = {"rank": None}

        def score_fn(idx, b):
            if bad_on["rank"] == rank:                             # only THIS rank's shard / member block meets the bad character
                raise ValueError("substring not found")
            return table_score(idx, b)

        ens = fd.DistributedEnsemble(members, mode=mode, score_fn=score_fn)
        calls = _count_collectives()
        out = ens.get_fitness(seqs)
        per_call = list(calls)
        del calls[:]
        mat = fd.DistributedEnsemble(members, mode=mode, score_fn=score_fn, combine_with=lambda x: x).get_fitness(seqs)
        per_matrix_call = list(calls)
        del calls[:]
        raised = []
        for bad_rank in (0, 1):
            bad_on["rank"] = bad_rank
            try:
                ens.get_fitness(seqs)
                raised.append(None)
            except ValueError as ex:
                raised.append(str(ex))
        bad_on["rank"] = None
        after = ens.get_fitness(seqs)                              # the flag does not stick to the next call
        q.put((rank, out, mat, per_call, per_matrix_call, raised, list(calls), after))
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["member", "sequence"])
def test_world2_get_fitness_is_one_collective_and_errors_reach_every_rank(mode):
    """SURVEY.md 8(e) / flexs/ensemble.py:54-59 sharded: exactly ONE exchange step per call.  Round 5 agreed on "a character outside
    the alphabet" with a second collective (all-reduce of a flag) and a blocking `.item()`; the flag now rides in the padding of the
    gathered block.  Counted here on gloo: one all-gather per `get_fitness`, nothing else -- and a ValueError raised by ONE rank's
    scorer still ends the call on EVERY rank (sequence_utils.py:46 raises for the whole batch), three collectives for three calls."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_one_collective_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs)
    rng = np.random.default_rng(0)
    want_mat = table_score([0, 1, 2], np.frombuffer(b"TGCA", np.uint8)[rng.integers(0, 4, (101, 8))])
    for rank, out, mat, per_call, per_matrix_call, raised, err_calls, after in results:
        assert per_call == ["all_gather"] and per_matrix_call == ["all_gather"]
        assert np.array_equal(mat, want_mat) and np.array_equal(out, np.mean(want_mat, axis=1)) and np.array_equal(after, out)
        assert raised == ["substring not found"] * 2                # whichever rank met it, both raise
        assert err_calls == ["all_gather"] * 3


# ---------------------------------------------------------------- cache-sharded neighbour search
class OracleLocalCache:
    """CPU stand-in for this rank's device key store (the C restatement of the K4 rule)."""

    def __init__(self, L):
        self.L = L
        self.rows = np.zeros((0, L), np.uint8)

    def __len__(self):
        return self.rows.shape[0]

    def append(self, rows):
        self.rows = np.concatenate([self.rows, rows.reshape(-1, self.L)])

    def min_dist(self, q, mode):
        from oracle import c_oracle

        return c_oracle.min_dist(q, self.rows, mode)


def _nam_inputs():
    rng = np.random.default_rng(4)
    base = rng.integers(0, 4, 14)
    pool = []
    for _ in range(700):
        s = base.copy()
        m = rng.random(14) < 0.12
        s[m] = rng.integers(0, 4, m.sum())
        if rng.random() < 0.3:
            s = np.roll(s, 1)
        pool.append("".join("UGCA"[i] for i in s))
    pool = list(dict.fromkeys(pool))
    table = {s: float(rng.random()) for s in pool}
    return pool, table


class TableLandscape(flexs_amd.Landscape):
    def __init__(self, table):
        super().__init__("table")
        self.table = table

    def _fitness_function(self, seqs):
        return np.array([self.table[str(s)] for s in seqs])


def _blend_f64(signal, noise, d, alpha_tab):
    a = alpha_tab[d]                                    # noisy_abstract_model.py:93-94 in float64
    return a * signal + (1 - a) * noise


def _nam_trace(nam, land, pool, table):
    np.random.seed(9)
    nam.train(pool[:33], np.array([table[s] for s in pool[:33]]))
    outs = [nam.get_fitness(pool[33 + 57 * i: 90 + 57 * i]) for i in range(4)]
    outs.append(nam.get_fitness(pool[10:140]))
    return np.concatenate(outs), land.cost, nam.cost, list(nam.cache), float(np.random.random())


def _cache_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)
        keys = rng.integers(65, 69, (501, 9)).astype(np.uint8)
        keys[100] = keys[7]                                         # duplicate: the earlier index must win
        queries = keys[rng.integers(0, 501, 80)].copy()
        mut = rng.random(queries.shape) < 0.15
        queries[mut] = rng.integers(65, 69, mut.sum())
        sc = fd.ShardedCache(9, local_factory=OracleLocalCache)
        empty = sc.min_dist(queries)
        got, sizes = [], []
        for lo, hi in ((0, 1), (1, 2), (2, 9), (9, 300), (300, 501)):   # odd block sizes: round-robin must stay aligned
            sc.append(keys[lo:hi])
            got.append([sc.min_dist(queries, mode) for mode in (0, 1)])
            sizes.append((len(sc), len(sc._local)))
        pool, table = _nam_inputs()
        land = TableLandscape(table)
        nam = fd.ShardedNoisyAbstractModel(land, 0.85, local_factory=OracleLocalCache, blend_fn=_blend_f64)
        q.put((rank, empty, got, sizes, _nam_trace(nam, land, pool, table)))
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_world2_sharded_cache_and_nam():
    from oracle import c_oracle, ref_np

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cache_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs)
    rng = np.random.default_rng(1)
    keys = rng.integers(65, 69, (501, 9)).astype(np.uint8)
    keys[100] = keys[7]
    queries = keys[rng.integers(0, 501, 80)].copy()
    mut = rng.random(queries.shape) < 0.15
    queries[mut] = rng.integers(65, 69, mut.sum())
    pool, table = _nam_inputs()
    land = TableLandscape(table)
    want_trace = _nam_trace(ref_np.NoisyAbstractModelOracle(land, 0.85), land, pool, table)
    for rank, empty, got, sizes, trace in results:
        assert (empty[0] == 0).all() and (empty[1] == -1).all()
        for (lo, hi), per_mode, (n_global, n_local) in zip(((0, 1), (1, 2), (2, 9), (9, 300), (300, 501)), got, sizes):
            assert n_global == hi and n_local == len(range(rank, hi, world))
            for mode, (d, a) in enumerate(per_mode):
                d_want, a_want = c_oracle.min_dist(queries, keys[:hi], mode)
                assert np.array_equal(d, d_want) and np.array_equal(a, a_want), (rank, hi, mode)
        assert np.array_equal(trace[0], want_trace[0]) and trace[1:] == want_trace[1:]


def test_sharded_cache_single_process():
    """No process group: world = 1, the sharded store degenerates to the local one."""
    from oracle import c_oracle

    rng = np.random.default_rng(2)
    keys = rng.integers(65, 69, (200, 7)).astype(np.uint8)
    sc = fd.ShardedCache(7, local_factory=OracleLocalCache)
    sc.append(keys[:50]); sc.append(keys[50:])
    d, a = sc.min_dist(keys[::3])
    d_want, a_want = c_oracle.min_dist(keys[::3], keys, 0)
    assert np.array_equal(d, d_want) and np.array_equal(a, a_want)


# ---------------------------------------------------------------- member-sharded training + the weight all-gather
def _train_inputs():
    rng = np.random.default_rng(3)
    seqs = ["".join("TGCA"[i] for i in row) for row in rng.integers(0, 4, (150, 8))]
    return seqs, rng.random(150)


def _make_trainables():
    from flexs_amd.baselines import models as bm

    # different architectures in one ensemble: the gathered blobs are padded to the largest member
    return [bm.MLP(8, 16, "TGCA", seed=0, epochs=2, batch_size=64), bm.GlobalEpistasisModel(8, 12, "TGCA", seed=1, epochs=2, batch_size=64),
            bm.CNN(8, 4, 8, "TGCA", kernel_size=3, seed=2, epochs=1, batch_size=64)]


def _train_worker(rank, world, port, q):
    import torch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs, y = _train_inputs()
        ens = fd.DistributedEnsemble(_make_trainables(), mode="member", score_fn=table_score)
        ens.train(seqs, y, seed=11)
        trained_here = [getattr(m.model, "_opt_state", None) is not None for m in ens.models]
        w_after_train = [m.model.get_weights() for m in ens.models]
        if rank == 1:                                               # rank 1 diverges; one flat broadcast realigns it
            ens.models[0].model.set_weights([w * 0 for w in ens.models[0].model.get_weights()])
        ens.broadcast_weights(src=0)
        q.put((rank, trained_here, w_after_train, [m.model.get_weights() for m in ens.models]))
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_world2_member_sharded_train_equals_single_process():
    """DistributedEnsemble.train in member mode: rank r trains only its member block, ONE all-gather of the weight blobs
    (flexs/ensemble.py:42-52 sharded as the scoring is); with per-member seeds the weights on every rank are those of the
    single-process Ensemble.train, bit for bit."""
    import torch

    torch.set_num_threads(1)
    seqs, y = _train_inputs()
    single = flexs_amd.Ensemble(_make_trainables())
    single.train(seqs, y, seed=11)
    want = [m.model.get_weights() for m in single.models]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for rank, trained_here, w_train, w_bcast in _collect(q, procs):
        assert trained_here == [i in fd.member_assignment(3, rank, world) for i in range(3)]
        for got_sets in (w_train, w_bcast):
            for got, ref in zip(got_sets, want):
                assert len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))


def _train_fail_worker(rank, world, port, q):
    import torch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs, y = _train_inputs()
        members = _make_trainables()
        members[2].model.loss = "huber"                              # member 2 lives on rank 1 only: fit raises ValueError THERE
        ens = fd.DistributedEnsemble(members, mode="member", score_fn=table_score)
        calls = _count_collectives()
        try:
            ens.train(seqs, y, seed=11)
            q.put((rank, "no exception", None, calls))
        except Exception as exc:                                      # noqa: BLE001
            q.put((rank, type(exc).__name__, str(exc), calls))
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_world2_member_sharded_train_failure_reaches_every_rank():
    """A failure that only the owning rank can see (an unsupported loss of its member) must end `train` on EVERY rank: the
    other rank used to wait forever in the weight all-gather (round-3 advisor finding)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_fail_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in _collect(q, procs)}
    assert got[1][0] == "ValueError" and "unsupported loss" in got[1][1]
    assert got[0][0] == "RuntimeError" and "another rank" in got[0][1]
    assert got[0][2] == ["all_gather"] and got[1][2] == ["all_gather"]      # the failure flag rode in the ONE weight gather


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run on 127.0.0.1); the
    launch path -- rendezvous, double-buffered launch / finish in both modes, MAX over ranks, ONE JSON line from rank 0
    -- is driven here on gloo with an injected scorer (`--cpu-selftest`)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-selftest"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                            # rank 0 only
    d = json.loads(lines[0])
    assert d["ranks"] == 2 and d["backend"] == "gloo" and all(v["ok"] for v in d["selftest"].values())
    # the launcher form of the contract still works: a mismatch between --gpus and the launcher's world size is an error
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-selftest"],
                       capture_output=True, text=True, timeout=120, cwd=root, env=dict(env, RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_without_devices_fails_after_spawning():
    """On a box with fewer devices than --gpus the ranks are spawned first and then name what is missing (here: no GPU)."""
    import subprocess
    import sys

    from flexs_amd import _native

    if _native.lib().fx_device_count() >= 2:
        pytest.skip("two GPUs visible: the real 2-rank run is tests/test_gpu_multirank.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0
    assert "devices needed" in r.stderr or "no HIP device visible" in r.stderr
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "elastic" in r.stderr   # it did spawn
