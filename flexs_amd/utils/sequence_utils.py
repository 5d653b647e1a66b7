"""Sequence helpers with the interface of flexs/utils/sequence_utils.py.

The two codecs on the scoring path -- string -> one-hot and one-hot -> string --
run on the GPU (fx_encode_onehot, fx_argmax_decode); batch variants are provided
because per-sequence calls are launch-latency bound.  The mutation helpers are
explorer-side host utilities, re-implemented here only so that code written
against `flexs.utils.sequence_utils` finds the same names.
"""
import random
from typing import List, Union

import numpy as np

from flexs_amd import _native

AAS = "ILVAGMFYWEDQNHCRKSTP"
"""str: Amino acid alphabet for proteins (length 20 - no stop codon)."""

RNAA = "UGCA"
"""str: RNA alphabet (4 base pairs)."""

DNAA = "TGCA"
"""str: DNA alphabet (4 base pairs)."""

BA = "01"
"""str: Binary alphabet '01'."""


def strings_to_one_hot(sequences, alphabet: str) -> np.ndarray:
    """Batch encode: N strings of equal length -> (N, L, A) float32 (what
    keras_model.py:70-75 feeds the network), one kernel launch."""
    seq_bytes = _native.sequences_to_bytes(sequences)
    if seq_bytes.shape[0] == 0 or seq_bytes.shape[1] == 0:
        return np.zeros((seq_bytes.shape[0], seq_bytes.shape[1], len(alphabet)), np.float32)
    return _native.Engine.get().encode_onehot(seq_bytes, _native.make_lut(alphabet), len(alphabet))


def string_to_one_hot(sequence: str, alphabet: str) -> np.ndarray:
    """(L, A) float64 one-hot of one sequence (sequence_utils.py:32-47).
    Raises ValueError for a character outside the alphabet, like `str.index`."""
    if len(sequence) == 0:
        return np.zeros((0, len(alphabet)))
    return strings_to_one_hot([sequence], alphabet)[0].astype(np.float64)


def one_hots_to_strings(one_hots, alphabet: str) -> List[str]:
    """Batch decode: (P, L, A) -> P strings, per-position first-maximum argmax."""
    x = np.asarray(one_hots, np.float64)
    if x.ndim != 3:
        raise ValueError("expected (P, L, A)")
    if x.shape[0] == 0 or x.shape[1] == 0:
        return ["" for _ in range(x.shape[0])]
    if x.shape[2] > len(alphabet):
        # np.argmax may select a column that has no character -> IndexError in the reference
        idx = np.argmax(x, axis=2)
        if idx.max() >= len(alphabet):
            raise IndexError("string index out of range")
    chars = _native.Engine.get().argmax_decode(x, alphabet)
    return [row.tobytes().decode("latin-1") for row in chars]


def one_hot_to_string(one_hot: Union[List[List[int]], np.ndarray], alphabet: str) -> str:
    """(L, A) one-hot / score matrix -> string (sequence_utils.py:50-66)."""
    x = np.asarray(one_hot, np.float64)
    return one_hots_to_strings(x[None], alphabet)[0]


def construct_mutant_from_sample(pwm_sample: np.ndarray, one_hot_base: np.ndarray) -> np.ndarray:
    """Overlay the sampled positions of `pwm_sample` on `one_hot_base` (sequence_utils.py:20-29)."""
    one_hot = np.array(one_hot_base, dtype=np.float64, copy=True)
    rows, cols = np.nonzero(pwm_sample)
    one_hot[rows, :] = 0
    one_hot[rows, cols] = 1
    return one_hot


def generate_single_mutants(wt: str, alphabet: str) -> List[str]:
    """Wild type followed by every single substitution, position-major (sequence_utils.py:69-77): len(wt) *
    len(alphabet) variants, the identity substitution included, every variant one edit away from `wt`."""
    sequences = [wt]
    for i in range(len(wt)):
        head, tail = wt[:i], wt[i + 1:]
        sequences.extend(head + ch + tail for ch in alphabet)
    return sequences


def generate_random_sequences(length: int, number: int, alphabet: str) -> List[str]:
    """`number` uniform random sequences (python `random`, sequence_utils.py:80-84)."""
    return ["".join([random.choice(alphabet) for _ in range(length)]) for _ in range(number)]


def generate_random_mutant(sequence: str, mu: float, alphabet: str) -> str:
    """Each residue is redrawn with probability `mu` (sequence_utils.py:87-108)."""
    # (same draws in the same order as the reference's loop -- one random() per residue, one choice() per redrawn residue;
    #  Adalead calls this a few thousand times per round, hence the local names and the comprehension)
    rnd, choice = random.random, random.choice
    return "".join([choice(alphabet) if rnd() < mu else s for s in sequence])
