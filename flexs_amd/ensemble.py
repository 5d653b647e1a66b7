"""`Ensemble` -- same contract as flexs/ensemble.py:10-59, with the member
forwards and the reduction fused into one engine call when every member is a
device surrogate."""
import os
from typing import Callable, List

import numpy as np

import flexs_amd
from flexs_amd import _native
from flexs_amd.types import SEQUENCES_TYPE


def _default_combine(x):
    return np.mean(x, axis=1)              # ensemble.py:24


def _device_members(models):
    """True if all members can be scored by one fused fx_score call."""
    from flexs_amd.baselines.models.keras_model import KerasModel

    if not models or not all(isinstance(m, KerasModel) for m in models):
        return False
    m0 = models[0]
    return all(m.model.L == m0.model.L and m.alphabet == m0.alphabet and m._device == m0._device for m in models)


if os.environ.get("FLEXS_AMD_BIND_FLEXS") == "1":
    import flexs as _flexs

    _EnsembleBase = _flexs.Ensemble
else:
    _EnsembleBase = flexs_amd.Model


def train_members(models, sequences, labels, seeds=None):
    """`for model in models: model.train(sequences, labels)` (flexs/ensemble.py:42-52, adaptive_ensemble.py:84-92,
    dyna_ppo.py:104-107).  When every member is a device surrogate with the stock `train`, the same loop with the members'
    training steps interleaved on the GPU (flexs_amd/training.py fit_many); any other member list is trained one by one.
    `seeds` (one per member, device surrogates only) fixes each member's shuffles and dropout masks: a member then ends
    up with the same weights whichever process trains it and whatever it is trained next to (member-sharded training)."""
    from flexs_amd.baselines.models.keras_model import KerasModel

    stock = [isinstance(m, KerasModel) and type(m).train is KerasModel.train for m in models]
    if len(models) > 1 and all(stock):
        from flexs_amd import training

        training.fit_many([m.model for m in models], sequences, labels, [m.alphabet for m in models],
                          [m.batch_size for m in models], [m.epochs for m in models], seeds=seeds)
        return
    for k, model in enumerate(models):
        if seeds is not None and stock[k]:
            model.train(sequences, labels, seed=seeds[k])
        else:
            model.train(sequences, labels)


class Ensemble(_EnsembleBase):
    """
    Ensemble of models / landscapes.

    Attributes:
        models: members.
        combine_with: (num_seqs, num_models) -> (num_seqs,) reduction; default mean.
    """

    def __init__(
        self,
        models: List[flexs_amd.Landscape],
        combine_with: Callable[[np.ndarray], np.ndarray] = _default_combine,
    ):
        name = f"Ens({'|'.join(model.name for model in models)})"            # ensemble.py:36
        flexs_amd.Model.__init__(self, name)
        self.models = models
        self.combine_with = combine_with

    def train(self, sequences: SEQUENCES_TYPE, labels: np.ndarray, seed: int = None):
        """ensemble.py:42-52.  `seed` (optional, not in the reference): member k trains with seed + k."""
        seeds = None if seed is None else [seed + k for k in range(len(self.models))]
        train_members(self.models, sequences, labels, seeds)

    _small = None                                          # explorer-size fast path: (members, fx_models, {want_mean: plan})

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_small", None)                           # device handles stay behind (as in KerasModel.__getstate__)
        return state

    def _score_small(self, sequences):
        """Calls of up to 4096 strings (SMALL_CALL_ROWS): string packing + fx_score in ONE C call (csrc/strpack.c score_small) on an
        argument block cached per member list.  None = not for this path (the general one below decides and raises).
        (Written as plain loops over cached tuples: at ~10 us per resident call every generator expression, bound-method call and
        dictionary lookup on the way is a visible share -- this method was 2.5 us, now ~1.5.)"""
        models = self.models
        c = self._small                                     # [members, fx_models (None: not device members), {want_mean: plan}, engine]
        same = c is not None and len(c[0]) == len(models)
        if same:
            for a, b in zip(c[0], models):
                if a is not b:
                    same = False
                    break
        if not same:
            dev = _device_members(models)
            c = self._small = [list(models), [m.native() for m in models] if dev else None, {}, models[0]._engine() if dev else None]
        natives = c[1]
        if natives is None:
            return None
        i = 0
        for m in models:                                    # (a member's new weights are uploaded by native(); same handle unless re-created)
            a = m.model
            nat = natives[i]
            if m._native_model is not nat or m._native_version != (id(a), getattr(a, "_version", 0)):
                nat = m.native()
                if nat is not natives[i]:
                    natives[i] = nat
                    c[2].clear()
            i += 1
        want_mean = self.combine_with is _default_combine
        plan = c[2].get(want_mean)
        if plan is None:
            m0 = models[0]
            plan = c[2][want_mean] = _native.small_plan(c[3], natives, m0.model.L, m0._lut, want_mean) or b""
        if not plan:
            return None
        n = len(sequences)
        out = np.empty(n if want_mean else (n, len(models)), np.float32)
        st = _native._strpack.score_small(plan, sequences, out)
        if st == 0:
            for m in models:                                # ensemble.py:55-57: every member's cost grows by N
                m.cost += n
            return out if want_mean else self.combine_with(out)
        if st == -1 or 1000 < st < 2000:
            return None                                     # (not for this path: the general one raises what the reference raises)
        for m in models:
            m.cost += n                                     # (the reference charges the members before a member's predict raises)
        _native.raise_small_status(st, c[3].handle)

    def _fitness_function(self, sequences):
        if (type(sequences) is np.ndarray and sequences.dtype.kind == "U" and sequences.ndim == 1
                and 0 < sequences.shape[0] <= _native.SMALL_CALL_ROWS):
            sequences = sequences.tolist()                      # (explorers also pass small NumPy string arrays: same fast path)
        if type(sequences) in (list, tuple) and 0 < len(sequences) <= _native.SMALL_CALL_ROWS and _native._HAS_SCORE_SMALL:
            out = self._score_small(sequences)
            if out is not None:
                return out
        if _device_members(self.models):
            # ensemble.py:55-57 calls member.get_fitness -> every member's cost grows by N
            n = len(sequences)
            for m in self.models:
                m.cost += n
            m0 = self.models[0]
            if _native.wants_chunked(sequences, m0.model.L):
                fused_mean = self.combine_with is _default_combine
                nm, mean = m0._engine().score_strings([m.native() for m in self.models], sequences, m0.model.L, m0._lut,
                                                      want_matrix=not fused_mean, want_mean=fused_mean)
                return mean if fused_mean else self.combine_with(nm)
            seq_bytes = _native.sequences_to_bytes(sequences, L=m0.model.L, staging=m0._engine())
            if seq_bytes.shape[0] == 0:
                scores = np.zeros((0, len(self.models)), np.float32)
                return self.combine_with(scores)
            natives = [m.native() for m in self.models]
            fused_mean = self.combine_with is _default_combine
            nm, mean = m0._engine().score(natives, seq_bytes, m0._lut, want_matrix=not fused_mean, want_mean=fused_mean)
            return mean if fused_mean else self.combine_with(nm)
        # heterogeneous / foreign members: the reference's host-side stacking
        # (foreign flexs.Model objects can only be called from Python; combine_with is user code)
        scores = np.stack([model.get_fitness(sequences) for model in self.models], axis=1)
        return self.combine_with(scores)
