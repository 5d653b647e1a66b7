"""Batched form of the explorers' decode-then-score inner step.

CMA-ES (`flexs/baselines/explorers/cmaes.py:61-67, 83-93`) and the DyNA-PPO environments
(`environments/dyna_ppo.py:144-163`) turn a float (L, A) array into a sequence by per-position argmax and ask
the model for ONE sequence at a time -- 15-40 launch-latency-bound `get_fitness([seq])` calls per iteration.
`PopulationEvaluator.evaluate` answers a whole population in one device round trip (`fx_decode_score`):
decode, score with every member, reduce, copy back.  What the explorer observes is unchanged:

* strings: `alphabet[argmax]` per position, first maximum wins -- the intermediate hard one-hot of
  `_soln_to_string` has the same argmax;
* values: a sequence found in one of the caller's `known` dicts (first dict wins, cmaes.py:86-89) gets that
  value; every other one gets `model.get_fitness([seq]).item()`'s value;
* `model.cost` (and each ensemble member's) grows by the number of sequences that were NOT known, exactly as
  the one-by-one loop charges them.

Models that are not device surrogates are called one sequence at a time like the reference does (their
answers may depend on call order, e.g. NoisyAbstractModel's cache and RNG).
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

from flexs_amd import _native
from flexs_amd.ensemble import Ensemble, _default_combine, _device_members


def _fused_members(model):
    from flexs_amd.baselines.models.keras_model import KerasModel

    if isinstance(model, KerasModel):
        return [model]
    if isinstance(model, Ensemble) and model.combine_with is _default_combine and _device_members(model.models):
        return list(model.models)
    return None


# Where the argmax of `_soln_to_string` runs for host arrays.  True: on the host (csrc/strpack.c decode_argmax, the rows split
# over the packing threads), and only P x L bytes go to the device -- a population of 40 237-residue solutions is 1.5 MB of
# float64, and staging that for the K6 kernel costs more than the scoring launch.  False: fx_decode_score (K6 + scoring + mean
# in one device round trip, the form of rounds 1-3; still what device-resident inputs use).  Same strings, same values.
HOST_DECODE = True


def _decode_host(x: np.ndarray, alphabet: str):
    """(P, L, A) float64 -> (P, L) uint8 characters, or None when the C helper is not built."""
    sp = _native._strpack
    if sp is None or not hasattr(sp, "decode_argmax"):
        return None
    x = np.ascontiguousarray(x, np.float64)
    P, L, A = x.shape
    out = np.empty((P, L), np.uint8)
    if sp.decode_argmax(x, P * L, A, alphabet.encode("latin-1"), out) != 0:
        return None
    return out


class PopulationEvaluator:
    def __init__(self, model, alphabet: str, seq_len: int):
        self.model = model
        self.alphabet = alphabet
        self.seq_len = seq_len
        self._members = _fused_members(model)
        self._plan_key, self._plan, self._alpha = None, b"", b""
        if self._members is not None and (self._members[0].alphabet != alphabet or self._members[0].model.L != seq_len):
            raise ValueError("PopulationEvaluator: alphabet / seq_len differ from the model's")

    def _as_one_hot(self, solutions) -> np.ndarray:
        x = np.asarray(solutions, np.float64)
        return x.reshape((-1, self.seq_len, len(self.alphabet)))              # cmaes.py:62

    def decode(self, solutions) -> List[str]:
        """`[_soln_to_string(s) for s in solutions]` on the device."""
        x = self._as_one_hot(solutions)
        if x.shape[0] == 0:
            return []
        chars = _decode_host(x, self.alphabet) if HOST_DECODE else None
        if chars is None:
            chars = _native.Engine.get(getattr(self.model, "_device", None)).argmax_decode(x, self.alphabet)
        return [r.tobytes().decode("latin-1") for r in chars]

    def evaluate(self, solutions, known: Sequence[Dict[str, float]] = ()) -> Tuple[List[str], np.ndarray]:
        x = self._as_one_hot(solutions)
        P = x.shape[0]
        values = np.empty(P, np.float64)
        if P == 0:
            return [], values
        if self._members is None:
            seqs = self.decode(x)
            for i, seq in enumerate(seqs):
                hit = next((d for d in known if seq in d), None)
                values[i] = hit[seq] if hit is not None else self.model.get_fitness([seq]).item()
            return seqs, values
        m0 = self._members[0]
        natives = [m.native() for m in self._members]
        single = len(natives) == 1 and self.model is m0
        step = self._one_call(x, natives, single) if HOST_DECODE else None
        if step is not None:
            seqs, scores = step
            return seqs, self._account(seqs, scores, known, values, single)
        chars = _decode_host(x, self.alphabet) if HOST_DECODE else None
        if chars is not None:
            nm, mean = m0._engine().score(natives, chars, m0._lut, want_matrix=single, want_mean=not single)
        else:
            chars, nm, mean = m0._engine().decode_score(natives, x, self.alphabet, m0._lut, want_matrix=single,
                                                        want_mean=not single)
        scores = nm[:, 0] if single else mean
        seqs = [r.tobytes().decode("latin-1") for r in chars]
        return seqs, self._account(seqs, scores, known, values, single)

    def _one_call(self, x: np.ndarray, natives, single: bool):
        """argmax + scoring + the rows as str in ONE C call (csrc/strpack.c population_step) on an argument block cached per member
        list; None when the helper is not built (the caller takes the step in pieces)."""
        sp = _native._strpack
        if sp is None or not hasattr(sp, "population_step"):
            return None
        key = (tuple(id(n) for n in natives), single)
        if self._plan_key != key:
            m0 = self._members[0]
            self._plan = _native.small_plan(m0._engine(), natives, self.seq_len, m0._lut, want_mean=not single) or b""
            self._plan_key = key
            self._alpha = self.alphabet.encode("latin-1")
        if not self._plan:
            return None
        x = np.ascontiguousarray(x, np.float64)
        P, L, A = x.shape
        chars = np.empty((P, L), np.uint8)
        out = np.empty((P, 1) if single else (P,), np.float32)
        st, seqs = sp.population_step(self._plan, x, P, A, self._alpha, chars, out)
        if st == 1:
            return None
        if st:
            _native.raise_small_status(st, self._members[0]._engine().handle)
        return seqs, (out[:, 0] if single else out)

    def _account(self, seqs, scores, known, values, single):
        fresh = 0
        if not known:
            values[:] = scores
            fresh = len(seqs)
            seqs = ()
        for i, seq in enumerate(seqs):
            hit = next((d for d in known if seq in d), None)
            if hit is not None:
                values[i] = hit[seq]
            else:
                values[i] = float(scores[i])
                fresh += 1
        self.model.cost += fresh                                               # landscape.py:44, one per unseen call
        if not single:
            for m in self._members:
                m.cost += fresh                                                # ensemble.py:55-57
        return values


def terminal_rewards(evaluator: PopulationEvaluator, seen, states: np.ndarray, lam: float):
    """Terminal branch of the DyNA-PPO environment step (environments/dyna_ppo.py:144-163) for the whole
    environment batch: decode `states[:, :, :-1]` (the last column is the mask token), score the sequences,
    record them in `seen` (a `flexs_amd.utils.edit_distance.SeenSequences`, the environment's `all_seqs`) and
    return `(sequences, fitnesses, rewards)` with reward = fitness - lam * sequence_density, the density being
    taken after the whole batch was recorded, as in the reference."""
    states = np.asarray(states, np.float64)
    seqs, fitnesses = evaluator.evaluate(states[:, :, :-1])
    if hasattr(seen, "add_many"):
        seen.add_many(seqs, fitnesses)                       # (one upload of the new keys)
    else:
        for seq, f in zip(seqs, fitnesses):
            seen.add(seq, f)
    dens = seen.densities(seqs)
    rewards = np.array([f - lam * d for f, d in zip(fitnesses, dens)])
    return seqs, fitnesses, rewards
