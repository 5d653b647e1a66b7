"""Multi-GPU scoring: one process per GPU, `torch.distributed` (backend "nccl"
== RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The hot path shards two natural ways, each with exactly
one exchange step -- an all-gather of float32 scores:

  member-parallel    rank r owns members {m : m % world == r}; every rank scores
                     the same N sequences with its members -> (N, M_r); ONE
                     all-gather of the padded (N, ceil(M/world)) blocks rebuilds
                     the stacked (N, M) matrix of ensemble.py:55-57 on every
                     rank; the reduction (np.mean order) then runs locally.
  sequence-parallel  every rank holds all members; rank r scores the contiguous
                     shard [lo_r, hi_r) of the batch with every member; ONE
                     all-gather of the padded (ceil(N/world), M) blocks (bench.py,
                     which only needs the default mean, gathers the per-rank means
                     instead: 4 bytes per sequence).

Messages are <= a few MB, i.e. latency-bound on xGMI (7 point-to-point links per
GPU), so a single collective per call is the design point -- no bucketing, no
ring pipelining.

`score_fn(member_indices, seq_bytes) -> (n, len(member_indices)) float32` is
injected so the sharding / gather logic is testable on CPU with gloo; the
default scores on this rank's GPU through the engine.

The NoisyAbstractModel neighbour search shards over the CACHE instead
(`ShardedCache`): global entry i lives on rank i % world, every rank runs K4
over its entries for all Q queries and ONE all-gather of Q 8-byte keys
`(remapped distance << 32 | global index)` followed by an element-wise min
reproduces the reference's "first entry at distance 1, else first strict
minimum" rule, because the global index encodes insertion order.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

import flexs_amd
from flexs_amd import _native
from flexs_amd.ensemble import _default_combine


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of `n` items for `rank` (first n % world shards get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def member_assignment(num_members: int, rank: int, world: int) -> List[int]:
    return list(range(rank, num_members, world))


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _gather_device(group):
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def all_gather_padded(local: np.ndarray, rows_max: int, group=None) -> List[np.ndarray]:
    """All-gather equal-shape float32 blocks (pad rows to `rows_max`), return the list of blocks."""
    rank, world = _world(group)
    block = np.zeros((rows_max,) + local.shape[1:], np.float32)
    block[: local.shape[0]] = local
    if world == 1:
        return [block]
    dev = _gather_device(group)
    send = torch.from_numpy(block).to(dev)
    recv = torch.empty((world,) + tuple(block.shape), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group) if dev.type == "cuda" else \
        dist.all_gather(list(recv.unbind(0)), send, group=group)
    out = recv.cpu().numpy()
    return [out[r] for r in range(world)]


class DistributedEnsemble(flexs_amd.Model):
    """`flexs.Ensemble` semantics across the GPUs of one node (SPMD: every rank
    calls `get_fitness` with the same sequences and receives the full result)."""

    def __init__(self, models: Sequence[flexs_amd.Model], mode: str = "member",
                 combine_with: Callable[[np.ndarray], np.ndarray] = _default_combine, group=None,
                 score_fn: Optional[Callable] = None):
        if mode not in ("member", "sequence"):
            raise ValueError("mode must be 'member' or 'sequence'")
        super().__init__(f"Ens({'|'.join(m.name for m in models)})")
        self.models = list(models)
        self.mode = mode
        self.combine_with = combine_with
        self.group = group
        self._score_fn = score_fn or self._score_on_engine

    # -- default scorer: this rank's GPU
    def _score_on_engine(self, member_idx: List[int], seq_bytes: np.ndarray) -> np.ndarray:
        if not member_idx or seq_bytes.shape[0] == 0:
            return np.zeros((seq_bytes.shape[0], len(member_idx)), np.float32)
        ms = [self.models[i] for i in member_idx]
        nm, _ = ms[0]._engine().score([m.native() for m in ms], seq_bytes, ms[0]._lut, want_matrix=True)
        return nm

    def train(self, sequences, labels):
        # every rank trains every member identically only if seeded identically; the usual
        # pattern is: train on rank 0, broadcast weights (see broadcast_weights)
        for m in self.models:
            m.train(sequences, labels)

    def broadcast_weights(self, src: int = 0):
        """Ship rank `src`'s member weights to every rank (after a round's `train`)."""
        rank, world = _world(self.group)
        if world == 1:
            return
        dev = _gather_device(self.group)
        for m in self.models:
            ws = m.model.get_weights()
            flat = torch.from_numpy(np.concatenate([w.ravel() for w in ws])).to(dev)
            dist.broadcast(flat, src=src, group=self.group)
            flat = flat.cpu().numpy()
            out, off = [], 0
            for w in ws:
                out.append(flat[off: off + w.size].reshape(w.shape))
                off += w.size
            m.model.set_weights(out)

    def _fitness_function(self, sequences):
        rank, world = _world(self.group)
        n, M = len(sequences), len(self.models)
        for m in self.models:
            m.cost += n                                            # ensemble.py:55-57 via landscape.py:44
        L = self.models[0].model.L if hasattr(self.models[0], "model") else None
        seq_bytes = _native.sequences_to_bytes(sequences, L=L)
        if self.mode == "member":
            mine = member_assignment(M, rank, world)
            per = -(-M // world)
            local = self._score_fn(mine, seq_bytes)                # (n, len(mine))
            blocks = all_gather_padded(np.ascontiguousarray(local.T), per, self.group)   # each (per, n)
            scores = np.empty((n, M), np.float32)
            for r in range(world):
                idx = member_assignment(M, r, world)
                scores[:, idx] = blocks[r][: len(idx)].T
        else:
            lo, hi = shard_range(n, rank, world)
            per = -(-n // world) if n else 0
            local = self._score_fn(list(range(M)), seq_bytes[lo:hi])           # (hi-lo, M)
            blocks = all_gather_padded(local, per, self.group)
            scores = np.empty((n, M), np.float32)
            for r in range(world):
                a, b = shard_range(n, r, world)
                scores[a:b] = blocks[r][: b - a]
        if self.combine_with is _default_combine and self._score_fn == self._score_on_engine and n:
            return _native.Engine.get(getattr(self.models[0], "_device", None)).ensemble_mean(scores)   # K3 on this GPU
        return self.combine_with(scores)


# ---------------------------------------------------------------------------
# NoisyAbstractModel: cache-sharded neighbour search
# ---------------------------------------------------------------------------
_NO_ENTRY = np.iinfo(np.int64).max


def _remap(dist_: np.ndarray) -> np.ndarray:
    """Distance 1 sorts first, then 0, then 2, 3, ... (noisy_abstract_model.py:53-58: a distance-1
    entry returns immediately, otherwise the first strict minimum wins)."""
    d = dist_.astype(np.int64)
    return np.where(d == 1, 0, np.where(d == 0, 1, d))


class ShardedCache:
    """Drop-in for `_native.NativeCache` whose rows are dealt round-robin to the ranks of `group`
    (SPMD: every rank appends the same rows and asks the same queries)."""

    def __init__(self, row_bytes: int, group=None, local_factory: Optional[Callable] = None, device: int = None):
        self.L = row_bytes
        self.group = group
        self.rank, self.world = _world(group)
        make = local_factory or (lambda L: _native.NativeCache(_native.Engine.get(device), L))
        self._local = make(row_bytes)
        self._size = 0                                   # global number of entries

    def __len__(self):
        return self._size

    def append(self, keys_u8: np.ndarray):
        k = np.ascontiguousarray(keys_u8, np.uint8)
        first = (self.rank - self._size) % self.world    # first row of this block that belongs to this rank
        self._local.append(k[first:: self.world])
        self._size += k.shape[0]

    def min_dist(self, queries: np.ndarray, mode: int = _native.FX_LEVENSHTEIN):
        q = np.ascontiguousarray(queries, np.uint8)
        Q = q.shape[0]
        if self._size == 0:                              # noisy_abstract_model.py:44-45
            return np.zeros(Q, np.int32), np.full(Q, -1, np.int64)
        keys = np.full(Q, _NO_ENTRY, np.int64)
        if len(self._local) and Q:
            d, a = self._local.min_dist(q, mode)
            keys = (_remap(d) << 32) | (a.astype(np.int64) * self.world + self.rank)
        if self.world > 1 and Q:
            dev = _gather_device(self.group)
            send = torch.from_numpy(keys).to(dev)
            recv = torch.empty((self.world, Q), dtype=torch.int64, device=dev)
            if dev.type == "cuda":
                dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
            else:
                dist.all_gather(list(recv.unbind(0)), send, group=self.group)
            keys = recv.min(dim=0).values.cpu().numpy()
        dp = keys >> 32
        return np.where(dp == 0, 1, np.where(dp == 1, 0, dp)).astype(np.int32), keys & 0xFFFFFFFF


def ShardedNoisyAbstractModel(landscape, signal_strength: float = 0.9, distance: str = "levenshtein", group=None,
                              local_factory: Optional[Callable] = None, blend_fn: Optional[Callable] = None,
                              device: int = None):
    """`NoisyAbstractModel` whose O(Q*C) neighbour search is split over the ranks of `group`; values,
    cache order, landscape cost and RNG stream are those of the single-GPU model on every rank.
    `local_factory` / `blend_fn` replace the two device steps (gloo tests on CPU)."""
    from flexs_amd.baselines.models.noisy_abstract_model import NoisyAbstractModel

    class _Sharded(NoisyAbstractModel):
        def _new_device_cache(self, row_bytes: int):
            return ShardedCache(row_bytes, group=group, local_factory=local_factory, device=device)

        def _blend(self, signal, noise, dist_, alpha_tab):
            if blend_fn is not None:
                return blend_fn(signal, noise, dist_, alpha_tab)
            return super()._blend(signal, noise, dist_, alpha_tab)

    return _Sharded(landscape, signal_strength, distance=distance, device=device)
