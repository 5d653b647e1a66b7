"""`AdditiveAAVPackaging` -- same contract as flexs/landscapes/additive_aav_packaging.py:25-119
(SURVEY.md section 8f-4): every (position, residue) carries a measured log2 enrichment, a sequence's raw
fitness is their sum, normalised by the best attainable sum.

The per-sequence Python loop over positions becomes one device pass: the single-substitution data are laid
out as an (L, n_residues + 1) float64 table (last column = 0 for residues a position has no entry for) and
`fx_table_additive` accumulates it in position order, so raw sums are bit-identical to the reference's
`+=` loop.  Normalisation, the Gaussian noise draw (global legacy NumPy RNG, one draw per sequence in batch
order -- also when `noise == 0`, exactly like `np.random.normal(scale=0)`) and the clamp at 0 stay on the host.

The measurement file (`AAV2_single_subs.json`: {position: {residue: {"log2_<phenotype>_v_wt": x,
"log2_packaging_v_wt": y, ...}}}) is not part of this repository, nor of the reference checkout this was
built against; pass `data_file`, set $FLEXS_AAV_SINGLE_SUBS, or install the reference package's data.
"""
import json
import os
from typing import Dict

import numpy as np

import flexs_amd
from flexs_amd import _native
from flexs_amd.types import SEQUENCES_TYPE

# AAV2 VP1 capsid protein (UniProt P03135), 735 residues -- public sequence data.
AAV2_WT = (
    "MAADGYLPDWLEDTLSEGIRQWWKLKPGPPPPKPAERHKDDSRGLVLPGYKYLGPFNGLDKGEPVNEADAAALEHDKAYDRQLDSGDNPYLKYNHADAEFQERLKEDTSFGGNLGRAVFQ"
    "AKKRVLEPLGLVEEPVKTAPGKKRPVEHSPVEPDSSSGTGKAGQQPARKRLNFGQTGDADSVPDPQPLGQPPAAPSGLGTNTMATGSGAPMADNNEGADGVGNSSGNWHCDSTWMGDRVI"
    "TTSTRTWALPTYNNHLYKQISSQSGASNDNHYFGYSTPWGYFDFNRFHCHFSPRDWQRLINNNWGFRPKRLNFKLFNIQVKEVTQNDGTTTIANNLTSTVQVFTDSEYQLPYVLGSAHQG"
    "CLPPFPADVFMVPQYGYLTLNNGSQAVGRSSFYCLEYFPSQMLRTGNNFTFSYTFEDVPFHSSYAHSQSLDRLMNPLIDQYLYYLSRTNTPSGTTTQSRLQFSQAGASDIRDQSRNWLPG"
    "PCYRQQRVSKTSADNNNSEYSWTGATKYHLNGRDSLVNPGPAMASHKDDEEKFFPQSGVLIFGKQGSEKTNVDIEKVMITDEEEIRTTNPVATEQYGSVSTNLQRGNRQAATADVNTQGV"
    "LPGMVWQDRDVYLQGPIWAKIPHTDGHFHPSPLMGGFGLKHPPPQILIKNTPVPANPSTTFSAAKFASFITQYSTGQVSVEIEWELQKENSKRWNPEIQYTSNYNKSVNVDFTVDTNGVY"
    "SEPRPIGTRYLTRNL"
)

PHENOTYPES = ("heart", "lung", "kidney", "liver", "blood", "spleen")
_VIABLE_PACKAGING = -6          # substitutions packaging worse than this never enter the best sequence (:88-91)


def _default_data_file() -> str:
    path = os.environ.get("FLEXS_AAV_SINGLE_SUBS")
    if path:
        return path
    import importlib.util

    spec = importlib.util.find_spec("flexs")
    if spec is not None and spec.submodule_search_locations:
        return os.path.join(list(spec.submodule_search_locations)[0], "landscapes", "data", "additive_aav_packaging",
                            "AAV2_single_subs.json")
    raise FileNotFoundError("AdditiveAAVPackaging: pass data_file= or set FLEXS_AAV_SINGLE_SUBS to AAV2_single_subs.json")


class AdditiveAAVPackaging(flexs_amd.Landscape):
    """Additive landscape from AAV2 single-substitution tissue-tropism measurements."""

    def __init__(self, phenotype: str = "heart", minimum_fitness_multiplier: float = 1, start: int = 0, end: int = 735,
                 noise: int = 0, data_file: str = None, device: int = None):
        super().__init__(f"AdditiveAAVPackaging_phenotype={phenotype}")      # additive_aav_packaging.py:56
        self.sequences = {}
        self.phenotype = f"log2_{phenotype}_v_wt"
        self.mfm = minimum_fitness_multiplier
        self.start, self.end, self.noise = start, end, noise
        self.wild_type = AAV2_WT[start:end]
        with open(data_file or _default_data_file()) as f:
            raw = json.load(f)
        self.data = {int(p): subs for p, subs in raw.items() if start <= int(p) < end}      # :66-76, file order
        self.top_seq, self.max_possible = self.compute_max_possible()
        self._device = device
        self._table = None

    def compute_max_possible(self):
        """Best residue per measured position among substitutions that still package (> -6), and the sum of
        their scores; a position with no such residue contributes "M" and -10 (:80-99)."""
        letters, total = [], 0
        for subs in self.data.values():
            best_aa, best = "M", -10
            for aa, scores in subs.items():
                if scores[self.phenotype] > best and scores["log2_packaging_v_wt"] > _VIABLE_PACKAGING:
                    best_aa, best = aa, scores[self.phenotype]
            letters.append(best_aa)
            total += best
        return "".join(letters), total

    # ---------------------------------------------------------------- device table
    def __getstate__(self):
        state = self.__dict__.copy()           # copy / pickle: the device table is rebuilt on first use
        state["_table"] = None
        return state

    def _native_table(self) -> "_native.NativeTable":
        if self._table is None:
            L = self.end - self.start
            residues = sorted({aa for subs in self.data.values() for aa in subs})
            if len(residues) > 255 or any(len(aa) != 1 or ord(aa) > 255 or ord(aa) == 0 for aa in residues):
                raise ValueError("AdditiveAAVPackaging: residue keys must be single one-byte characters")
            zero = len(residues)
            lut = np.full(256, zero, np.uint8)
            for col, aa in enumerate(residues):
                lut[ord(aa)] = col
            table = np.zeros((L, zero + 1), np.float64)
            for pos, subs in self.data.items():
                for aa, scores in subs.items():
                    table[pos - self.start, lut[ord(aa)]] = scores[self.phenotype]
            self._measured = np.zeros(L, bool)
            self._measured[[p - self.start for p in self.data]] = True
            self._table = _native.NativeTable(_native.Engine.get(self._device), table, "", lut=lut)
        return self._table

    def _rows(self, sequences) -> np.ndarray:
        """(N, L) byte rows, NUL-padded: a short sequence simply sums fewer positions (:103-105); characters that
        cannot be residue keys hit the zero column like any other unmeasured residue."""
        L = self.end - self.start
        if self._measured.all():
            try:
                return _native.sequences_to_bytes(sequences, L=L)          # usual case: N full-length sequences
            except ValueError:
                pass
        rows = np.zeros((len(sequences), L), np.uint8)
        for n, seq in enumerate(sequences):
            seq = str(seq)
            for i in range(len(seq)):
                # the reference indexes self.data[start + i] and fails on the first position it has no data for
                if i >= L or not self._measured[i]:
                    raise KeyError(self.start + i)
            if seq:
                rows[n, : len(seq)] = np.frombuffer(seq.encode("latin-1", "replace"), np.uint8)
        return rows

    def _get_raw_fitness(self, seq) -> float:
        table = self._native_table()
        return float(table.additive_sum(self._rows([seq]))[0]) + self.mfm * self.max_possible

    def _fitness_function(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
        n = len(sequences)
        if n == 0:
            return np.array([])
        table = self._native_table()
        raw = table.additive_sum(self._rows(sequences)) + self.mfm * self.max_possible
        normed = raw / (self.max_possible * (self.mfm + 1))                     # :112-114
        noisy = normed + np.random.normal(scale=self.noise, size=n)             # one legacy-RNG draw per sequence
        if not (noisy > 0).any():
            return np.zeros(n, dtype=int)                                       # np.array([0, 0, ...]) of Python ints
        return np.where(noisy > 0, noisy, 0.0)                                  # max(0, x): NaN and -0.0 -> 0


def registry() -> Dict[str, Dict]:
    """Problem registry in the reference's format (:122-147): one problem per tissue on residues 450-540."""
    return {p: {"params": {"phenotype": p, "start": 450, "end": 540}} for p in PHENOTYPES}
