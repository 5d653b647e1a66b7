"""Multi-GPU scoring: one process per GPU, `torch.distributed` (backend "nccl"
== RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The hot path shards two natural ways, each with exactly
one exchange step -- an all-gather of float32 scores:

  member-parallel    rank r owns the contiguous member block [r*per, (r+1)*per), per = ceil(M/world); every rank scores
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
    """Members of rank `rank`: the contiguous block [rank * per, (rank + 1) * per), per = ceil(M / world) -- so the
    gathered member-major planes of all ranks ARE the stacked predictions in member order (no re-assembly pass)."""
    per = -(-num_members // world)
    return list(range(min(rank * per, num_members), min((rank + 1) * per, num_members)))


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _on_gpu(group) -> bool:
    """True when the exchange runs on device buffers (RCCL group, or no group at all on a GPU box)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_backend(group) == "nccl"
    return torch.cuda.is_available()


def _gather_device(group):
    return torch.device("cuda", torch.cuda.current_device()) if _on_gpu(group) else torch.device("cpu")


def _all_gather(recv: torch.Tensor, send: torch.Tensor, group=None, force: bool = False):
    """recv[r] = rank r's `send` (equal shapes).  The ONE collective of the path; on RCCL it is a single
    all_gather_into_tensor over device buffers, on gloo (CPU tests) the list form of the same collective."""
    if _world(group)[1] == 1 and not (force and dist.is_available() and dist.is_initialized()):
        recv[0].copy_(send)
    elif recv.is_cuda:
        dist.all_gather_into_tensor(recv.view(-1), send.reshape(-1), group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), send, group=group)


def _stride_for(n: int) -> int:
    """Plane stride in floats: 256-byte aligned planes plus one 64-float block of padding that no kernel writes.  Its last element
    on plane 0 of a rank's block carries that rank's error word through the all-gather (`_FLAG`)."""
    return (n + 63) // 64 * 64 + 64


_FLAG = -1          # s.local[0, _FLAG]: this rank's deferred error bits as a float (0.0 = none); s.recv[r, 0, _FLAG]: rank r's


class _Slot:
    """One in-flight call's buffers (bench.py double-buffers two of them so that the all-gather of step k runs on the
    communication stream while step k + 1 computes)."""

    def __init__(self):
        self.key = None
        self.planes = self.local = self.recv = self.mean = None
        self.done = None            # event on the communication stream: this slot's gather has landed
        self.exchange = False       # False with one rank: recv aliases local
        self.n = 0
        self.want = "mean"
        self.busy = False
        self.keep = None
        self.err = None             # (injected scorer only) the ValueError this rank's scorer raised


class DistributedEnsemble(flexs_amd.Model):
    """`flexs.Ensemble` semantics (flexs/ensemble.py:54-59) across the GPUs of one node.  SPMD: every rank calls
    `get_fitness` with the same sequences and receives the full result.

    The data path is device-resident.  Every rank uploads (its share of) the byte batch once over its own PCIe link,
    scores with the fused HIP kernels into member-major planes in HBM (`fx_score_planes_dev`), ONE all-gather over
    device buffers on a separate communication stream rebuilds the result on every rank, and only the final `(N,)`
    vector (or the `(N, M)` matrix for a custom `combine_with`) crosses back to the host:

      member mode    rank r scores its member block for all N rows; the gathered planes ARE the stacked predictions
                     (plane p = member p); the NumPy-order mean kernel then runs on them on every rank.
      sequence mode  rank r scores rows [lo_r, hi_r) with every member; for the default mean it reduces locally
                     first and gathers 4 bytes per sequence, else it gathers its (M, rows) planes.

    `launch` / `finish` are the two halves of a call, so that a caller with a stream of batches (bench.py) overlaps
    the gather of batch k with the scoring of batch k + 1; `get_fitness` is `launch` + `finish`.

    `score_fn(member_indices, seq_bytes) -> (n, len(member_indices)) float32` replaces the on-engine scorer (gloo
    tests on CPU); everything else -- assignment, padding, the collective, re-assembly, the reduction order -- is
    the same code on both backends."""

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
        self._score_fn = score_fn
        self._cuda = score_fn is None and torch.cuda.is_available()
        if score_fn is None and not self._cuda:
            raise RuntimeError("DistributedEnsemble scores on this rank's MI355X: no GPU is visible (there is no CPU "
                               "fallback); pass score_fn for CPU tests")
        # a non-RCCL group (gloo) with on-GPU scoring: the planes are staged through host tensors for the collective
        self._host_exchange = self._cuda and not _on_gpu(group)
        self._slots = [_Slot(), _Slot()]
        self._comm = None
        # with one rank the gathered block IS the local block (no copy, no second stream); tests and
        # `bench.py --force-dist` set this to run the real collective on a one-rank RCCL group
        self.force_collective = False

    # ------------------------------------------------------------------ plumbing
    def _engine(self):
        return _native.Engine.get(getattr(self.models[0], "_device", None))

    @property
    def stream(self):
        """Compute stream of the device path: the torch stream the scoring engine enqueues on."""
        return self._engine().torch_stream() if self._cuda else None

    def _shape(self, n: int, want: str):
        """(members scored here, rows scored here [lo, hi), planes gathered per rank, plane stride) for a batch."""
        rank, world = _world(self.group)
        M = len(self.models)
        if self.mode == "member":
            return member_assignment(M, rank, world), (0, n), -(-M // world), _stride_for(n)
        lo, hi = shard_range(n, rank, world)
        return list(range(M)), (lo, hi), (1 if want == "mean" else M), _stride_for(-(-n // world) if n else 0)

    # ------------------------------------------------------------------ training (flexs/ensemble.py:42-52)
    def _device_surrogates(self) -> bool:
        from flexs_amd.baselines.models.keras_model import KerasModel

        return all(isinstance(m, KerasModel) and type(m).train is KerasModel.train for m in self.models)

    def train(self, sequences, labels, seed: Optional[int] = None):
        """`for model in self.models: model.train(sequences, labels)` (ensemble.py:42-52, called once per explorer round,
        explorer.py:157-160), sharded like the scoring: in member mode rank r trains ONLY its member block
        (`member_assignment`: ceil(M / world) members per rank, interleaved on the GPU by `training.fit_many`), then ONE
        all-gather of the ranks' concatenated weight blobs puts every member's new weights on every rank
        (`gather_weights`).  Member k trains with `seed + k` wherever it is trained, so the result equals the
        single-process `Ensemble.train(..., seed=seed)` bit for bit; without a seed every member draws fresh shuffles /
        dropout masks, as Keras does.  The optimiser state of a member stays with the rank that owns it (the assignment
        is static).  Sequence mode (every rank scores with every member) and foreign members train everywhere."""
        from flexs_amd.ensemble import train_members

        rank, world = _world(self.group)
        seeds = None if seed is None else [seed + k for k in range(len(self.models))]
        if self.mode == "member" and world > 1 and self._device_surrogates():
            # every rank checks the whole training set first (ValueError of the reference's encode loop), so that a rank
            # that owns no member fails WITH the others instead of waiting for them in the weight gather
            for m in {(mm.alphabet, mm.model.L): mm for mm in self.models}.values():
                if len(sequences) and (m._lut[_native.sequences_to_bytes(sequences, L=m.model.L)] == 255).any():
                    raise ValueError("substring not found")
            mine = member_assignment(len(self.models), rank, world)
            # SPMD: whatever fails on the rank that owns a member (an unsupported loss, an out-of-memory arena, an exception
            # of a member's own `train`) must fail the call on EVERY rank -- the others would otherwise wait for it forever
            # in the weight gather below.  The failing rank still joins the gather; its flag travels in the block it sends.
            err = None
            try:
                train_members([self.models[i] for i in mine], sequences, labels,
                              None if seeds is None else [seeds[i] for i in mine])
            except Exception as ex:                                # noqa: BLE001 (re-raised below, on every rank)
                err = ex
            self.gather_weights(failed=err)                        # (raises on every rank if any rank failed)
        else:
            train_members(self.models, sequences, labels, seeds)

    def _blob_len(self) -> int:
        return max(m.model.count_params() for m in self.models)

    def _pack_weights(self, members: List[int], per: int, width: int) -> torch.Tensor:
        """(per, width) float32: row j = member members[j]'s weight arrays back to back (Keras order), zero padded."""
        host = np.zeros((per, width), np.float32)
        for j, i in enumerate(members):
            flat = np.concatenate([w.ravel() for w in self.models[i].model._weights])
            host[j, : flat.size] = flat
        return torch.from_numpy(host)

    def _unpack_weights(self, row: np.ndarray, i: int):
        arch, out, off = self.models[i].model, [], 0
        for shp in arch.shapes():
            size = int(np.prod(shp))
            out.append(row[off: off + size].reshape(shp))
            off += size
        arch.set_weights(out)

    def gather_weights(self, failed: Optional[BaseException] = None):
        """After a member-sharded `train`: ONE all-gather of every rank's `(ceil(M / world), P + 1)` weight block (P = the
        largest member's parameter count; the extra column of row 0 = 1.0 when this rank's training `failed`) over the gather
        device (device buffers on RCCL), then `set_weights` for the members this rank does not own.  8 x 22 429 floats =
        0.7 MB for the canonical CNN ensemble.  A failure on any rank raises on every rank -- the failing one its own
        exception, the others a RuntimeError -- and leaves the members' weights as they were."""
        rank, world = _world(self.group)
        if world == 1:
            if failed is not None:
                raise failed
            return
        M, dev = len(self.models), _gather_device(self.group)
        per, width = -(-M // world), self._blob_len()
        block = self._pack_weights(member_assignment(M, rank, world), per, width + 1)
        block[0, width] = 0.0 if failed is None else 1.0
        send = block.to(dev)
        recv = torch.empty((world, per, width + 1), dtype=torch.float32, device=dev)
        _all_gather(recv, send, self.group)
        got = recv.cpu().numpy()
        if failed is not None:
            raise failed
        if (got[:, 0, width] != 0).any():
            raise RuntimeError("DistributedEnsemble.train: training failed on another rank (see that rank's exception)")
        rows = got[:, :, :width].reshape(world * per, width)
        mine = set(member_assignment(M, rank, world))
        for i in range(M):
            if i not in mine:                                      # contiguous assignment: row i IS member i
                self._unpack_weights(rows[i], i)

    def broadcast_weights(self, src: int = 0):
        """Ship rank `src`'s member weights to every rank: ONE broadcast of all members' weights in one flat buffer."""
        rank, world = _world(self.group)
        if world == 1:
            return
        M, dev = len(self.models), _gather_device(self.group)
        flat = self._pack_weights(list(range(M)), M, self._blob_len()).to(dev)
        dist.broadcast(flat, src=src, group=self.group)
        if rank != src:
            rows = flat.cpu().numpy()
            for i in range(M):
                self._unpack_weights(rows[i], i)

    def _reduce_planes(self, planes: torch.Tensor, rows: int, M: int, stride: int, out: torch.Tensor):
        """out[:rows] = np.mean over the M member planes, NumPy float32 summation order (K3 on the device path)."""
        if rows == 0:
            return
        if self._cuda and M <= 16:
            self._engine().ensemble_mean_planes_dev(planes.data_ptr(), rows, M, stride, out.data_ptr())
        elif self._cuda:
            mat = planes[:M, :rows].t().contiguous()
            self._engine().ensemble_reduce_dev(mat.data_ptr(), rows, M, out.data_ptr())   # (mat is freed stream-ordered)
        else:
            out[:rows] = torch.from_numpy(_default_combine(planes[:M, :rows].t().contiguous().numpy()).astype(np.float32, copy=False))

    # ------------------------------------------------------------------ the two halves of one call
    def launch(self, seq, n: Optional[int] = None, slot: int = 0, want: str = "mean", timing=None):
        """Score this rank's share of the batch and start the all-gather.  `seq`: (n, L) uint8 -- a NumPy array (this
        rank's rows are uploaded), or on the device path a CUDA tensor of the whole batch already resident in HBM.
        want = "mean" | "matrix" (what `finish` will hand out); timing = (start, stop) torch.cuda.Event pair
        recorded on the compute stream around the scoring kernel (bench.py's kernel_ms)."""
        rank, world = _world(self.group)
        n = int(seq.shape[0]) if n is None else n
        M = len(self.models)
        mine, (lo, hi), per, stride = self._shape(n, want)
        s = self._slots[slot]
        dev = torch.device("cuda", torch.cuda.current_device()) if self._cuda else torch.device("cpu")
        key = (n, per, stride, world, want)
        if s.key != key:
            s.local = torch.zeros((per, stride), dtype=torch.float32, device=dev)
            s.exchange = world > 1 or (self.force_collective and dist.is_available() and dist.is_initialized())
            s.recv = torch.zeros((world, per, stride), dtype=torch.float32, device=dev) if s.exchange else s.local.unsqueeze(0)
            s.mean = torch.empty((max(n, 1),), dtype=torch.float32, device=dev)
            # sequence mode, mean only: the (M, rows) planes stay local, their mean is what travels
            local_reduce = self.mode == "sequence" and want == "mean"
            s.planes = torch.zeros((M, stride), dtype=torch.float32, device=dev) if local_reduce else s.local
            s.done = torch.cuda.Event() if self._cuda else None
            s.key, s.busy = key, False
            if self._cuda:
                torch.cuda.current_stream().synchronize()            # the zero fills ran on the caller's stream
        s.n, s.want = n, want
        local_reduce = s.planes is not s.local
        if self._cuda:
            m0 = self.models[0]
            st = self.stream
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            with torch.cuda.stream(st):
                if s.busy and s.exchange and not self._host_exchange:
                    st.wait_event(s.done)                            # the slot's previous gather has been consumed
                if isinstance(seq, np.ndarray):
                    seq = torch.from_numpy(np.ascontiguousarray(seq[lo:hi])).to(dev, non_blocking=True)
                    row0 = 0
                else:
                    row0 = lo
                if mine and hi > lo:
                    natives = [self.models[i].native() for i in mine]
                    if timing:
                        timing[0].record(st)
                    if local_reduce and M <= 16:
                        # scores + their NumPy-order mean in one call: the scoring kernel averages a tile itself where it can
                        # (the last member to finish it), otherwise the mean kernel follows inside the call
                        self._engine().score_mean_planes_dev(natives, seq.data_ptr() + row0 * m0.model.L, hi - lo, m0.model.L,
                                                             m0._lut, s.planes.data_ptr(), stride, s.local.data_ptr())
                        if timing:
                            timing[1].record(st)
                    else:
                        self._engine().score_planes_dev(natives, seq.data_ptr() + row0 * m0.model.L, hi - lo, m0.model.L,
                                                        m0._lut, s.planes.data_ptr(), stride)
                        if timing:
                            timing[1].record(st)
                        if local_reduce:
                            self._reduce_planes(s.planes, hi - lo, M, stride, s.local)
                if s.exchange:
                    # the error word travels WITH the scores: a character outside the alphabet fails the call on every rank
                    # (sequence_utils.py:46 raises for the whole batch) without a second collective or a host sync
                    self._engine().error_word_dev(s.local.data_ptr() + 4 * (stride - 1))
                s.keep = seq                                         # alive until the kernels have run
                if s.exchange and self._host_exchange:
                    host = s.local.cpu()                             # (stream-ordered, synchronous D2H)
                    got = torch.empty((world,) + tuple(host.shape), dtype=torch.float32)
                    _all_gather(got, host, self.group, force=True)   # gloo: the list form on host tensors
                    s.recv.copy_(got)
                elif s.exchange:
                    self._comm.wait_stream(st)
                    with torch.cuda.stream(self._comm):
                        _all_gather(s.recv, s.local, self.group, force=True)   # RCCL over xGMI, device buffers
                        s.done.record(self._comm)
        else:
            b = np.ascontiguousarray(seq[lo:hi])
            s.err = None
            try:
                local = self._score_fn(mine, b) if (mine and hi > lo) else np.zeros((hi - lo, len(mine)), np.float32)
            except ValueError as ex:                                 # the injected scorer's "character outside the alphabet"
                s.err, local = ex, np.zeros((hi - lo, len(mine)), np.float32)
            s.planes.zero_()
            s.planes[: len(mine), : hi - lo] = torch.from_numpy(np.ascontiguousarray(np.asarray(local, np.float32).T))
            if local_reduce:
                self._reduce_planes(s.planes, hi - lo, M, stride, s.local[0])
            s.local[0, _FLAG] = 0.0 if s.err is None else 1.0
            if s.exchange:
                _all_gather(s.recv, s.local, self.group)
        s.busy = True
        return s

    def finish(self, slot: int = 0):
        """Wait for the slot's gather and hand out what `launch` was asked for: the `(n,)` np.mean-order mean, or the
        stacked `(n, M)` predictions (`np.stack(axis=1)` of ensemble.py:55-57) -- on every rank, as a tensor on the
        gather device (CUDA on RCCL); the caller decides when to copy it to the host."""
        rank, world = _world(self.group)
        s = self._slots[slot]
        n, M, want = s.n, len(self.models), s.want
        _, per, stride, _, _ = s.key
        if self._cuda and s.exchange and not self._host_exchange:
            self.stream.wait_event(s.done)
        s.busy = False
        with (torch.cuda.stream(self.stream) if self._cuda else _Null()):
            if self.mode == "member":
                planes = s.recv.view(world * per, stride)            # plane p = member p (contiguous assignment)
                if want == "matrix":
                    return planes[:M, :n].t().contiguous()
                self._reduce_planes(planes, n, M, stride, s.mean)
                return s.mean[:n]
            spans = [shard_range(n, r, world) for r in range(world)]
            if want == "matrix":                                     # rank r's block: rows [lo_r, hi_r) of every member
                return torch.cat([s.recv[r, :, : b - a].t() for r, (a, b) in enumerate(spans)], dim=0).contiguous()
            if world == 1 or (n % world == 0 and n // world == stride):
                return s.recv.view(-1)[:n]                           # the shards already sit back to back
            return torch.cat([s.recv[r, 0, : b - a] for r, (a, b) in enumerate(spans)], dim=0)

    # ------------------------------------------------------------------ flexs.Model API
    def _fitness_function(self, sequences):
        n, M = len(sequences), len(self.models)
        for m in self.models:
            m.cost += n                                            # ensemble.py:55-57 via landscape.py:44
        L = self.models[0].model.L if hasattr(self.models[0], "model") else None
        seq_bytes = _native.sequences_to_bytes(sequences, L=L)
        if n == 0:
            return self.combine_with(np.zeros((0, M), np.float32))
        default = self.combine_with is _default_combine
        s = self.launch(seq_bytes, n, 0, "mean" if default else "matrix")
        out = self.finish(0)
        rank, world = _world(self.group)
        err, flags = s.err, None
        if self._cuda:
            with torch.cuda.stream(self.stream):
                if s.exchange:
                    # ONE stream-ordered D2H: the result with every rank's error word behind it (they came with the gather)
                    shape = tuple(out.shape)
                    both = torch.cat([out.reshape(-1), s.recv[:, 0, _FLAG]]).cpu()
                    out, flags = both[: both.numel() - world].reshape(shape), both[both.numel() - world:]
                else:
                    out = out.cpu()                                # stream-ordered D2H of the final result only
            try:
                self._engine().sync()                              # raises ValueError for a character outside the alphabet
            except ValueError as ex:
                err = ex
        elif s.exchange:
            flags = s.recv[:, 0, _FLAG]
        # SPMD: a character outside the alphabet fails the call on EVERY rank (sequence_utils.py:46 raises for the whole batch),
        # also on the ranks whose row shard / member block never met it
        if err is None and flags is not None and bool((flags != 0).any()):
            err = ValueError("substring not found")
        if err is not None:
            raise err
        out = out.numpy()
        return out if default else self.combine_with(out)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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
