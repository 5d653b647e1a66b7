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
    assert fd.member_assignment(8, 1, 8) == [1] and fd.member_assignment(3, 1, 2) == [1]
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
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,M,n", [("member", 3, 101), ("member", 8, 64), ("sequence", 3, 101), ("sequence", 2, 1),
                                      ("member", 1, 5)])
def test_world2_gloo(mode, M, n):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, M, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    seq_bytes = np.frombuffer(b"TGCA", np.uint8)[rng.integers(0, 4, (n, 8))]
    want_mat = table_score(list(range(M)), seq_bytes)
    for rank, out, mat, called, cost, mcosts in results:
        assert np.array_equal(mat, want_mat)                       # every rank holds the full stacked matrix
        assert np.array_equal(out, np.mean(want_mat, axis=1))
        assert cost == 2 * n // 2 and mcosts == [2 * n] * M        # two get_fitness calls on members, one per ensemble
        idx, rows = called[0]
        if mode == "member":
            assert idx == fd.member_assignment(M, rank, world) and rows == n
        else:
            lo, hi = fd.shard_range(n, rank, world)
            assert idx == list(range(M)) and rows == hi - lo
