"""`TFBinding` -- same contract as flexs/landscapes/tf_binding.py:12-93, with the
dict look-up replaced by a device-resident table indexed by the 2-bit-packed 8-mer.

The on-disk format is the reference's (Barrera et al. 2016 8-mer files: tab
separated `8-mer, 8-mer(.1), E-score, Median, Z-score`); the data files themselves
are not part of this repository -- point `landscape_file` (or `registry(data_dir)`)
at the reference's `flexs/landscapes/data/tf_binding/`.
"""
import os
from typing import Dict

import numpy as np
import pandas as pd

import flexs_amd
from flexs_amd import _native
from flexs_amd.types import SEQUENCES_TYPE

_ALPHABET = "ACGT"


class TFBinding(flexs_amd.Landscape):
    """Binding affinity of every DNA 8-mer to one transcription factor, normalised to [0, 1]."""

    batch_safe = True      # deterministic table: NoisyAbstractModel may query it in batches

    def __init__(self, landscape_file: str, device: int = None):
        super().__init__(name="TF_Binding")                          # tf_binding.py:29
        data = pd.read_csv(landscape_file, sep="\t")
        score = data["E-score"]
        norm_score = (score - score.min()) / (score.max() - score.min())   # tf_binding.py:33-34
        # both strands map to the same score; the second column overrides the first (dict.update order, :38-41)
        self.sequences = dict(zip(data["8-mer"], norm_score))
        self.sequences.update(zip(data["8-mer.1"], norm_score))
        self._L = len(next(iter(self.sequences)))
        self._device = device
        self._table = None

    def __getstate__(self):
        state = self.__dict__.copy()           # copy / pickle: the device table is rebuilt on first use
        state["_table"] = None
        return state

    def _native_table(self):
        if self._table is None:
            L = self._L
            table = np.full(4 ** L, np.nan)
            keys = list(self.sequences)
            kb = _native.sequences_to_bytes(keys, L=L)
            lut = _native.make_lut(_ALPHABET)
            codes = lut[kb].astype(np.int64)
            if (codes == 0xFF).any():
                raise ValueError("TFBinding file contains characters outside ACGT")
            idx = (codes << (2 * np.arange(L))).sum(axis=1)
            table[idx] = np.fromiter(self.sequences.values(), dtype=np.float64, count=len(keys))
            self._table = _native.NativeTable(_native.Engine.get(self._device), table, _ALPHABET, bits=2)
        return self._table

    def _fitness_function(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
        """tf_binding.py:43-44: np.array([self.sequences[seq] for seq in sequences]) (KeyError if absent)."""
        if len(sequences) == 0:
            return np.array([])
        try:
            seq_bytes = _native.sequences_to_bytes(sequences, L=self._L)
        except ValueError:
            seq_bytes = None
        if seq_bytes is None:
            return np.array([self.sequences[seq] for seq in sequences])   # raises the reference's KeyError
        out = self._native_table().lookup(seq_bytes)
        if np.isnan(out).any():
            bad = int(np.flatnonzero(np.isnan(out))[0])
            raise KeyError(str(sequences[bad]))
        return out


def registry(data_dir: str = None) -> Dict[str, Dict]:
    """Problem registry in the reference's format (tf_binding.py:47-93).  `data_dir` defaults to
    $FLEXS_TF_BINDING_DIR, then to the installed reference package's data directory."""
    if data_dir is None:
        data_dir = os.environ.get("FLEXS_TF_BINDING_DIR")
    if data_dir is None:
        import importlib.util

        spec = importlib.util.find_spec("flexs")
        if spec is None or not spec.submodule_search_locations:
            raise FileNotFoundError("set FLEXS_TF_BINDING_DIR to the directory holding the *_8mers.txt files")
        data_dir = os.path.join(list(spec.submodule_search_locations)[0], "landscapes", "data", "tf_binding")
    starts = ["GCTCGAGC", "GCGCGCGC", "TGCGCGCC", "ATATAGCC", "GTTTGGTA", "ATTATGTT", "CAGTTTTT",
              "AAAAATTT", "AAAAACGC", "GTTGTTTT", "TGCTTTTT", "AAAGATAG", "CCTTCTTT", "AAAGAGAG"]
    problems = {}
    for fname in os.listdir(data_dir):
        problems[fname.replace("_8mers.txt", "")] = {
            "params": {"landscape_file": os.path.join(data_dir, fname)},
            "starts": list(starts),
        }
    return problems
