"""Edit-distance services built on the K4 kernels.

`SeenSequences` is the device-backed form of the `all_seqs` bookkeeping in the
DyNA-PPO environments (flexs/baselines/explorers/environments/dyna_ppo.py:106-114,
267-275): `density(seq)` = sum of fitness / distance over every observed sequence
at edit distance 1..radius, accumulated in insertion order like the reference loop
(so the float result is bit-identical)."""
import numpy as np

from flexs_amd import _native


class SeenSequences:
    def __init__(self, seq_len: int, distance: str = "levenshtein", device: int = None):
        """`seq_len` = row width on the device = the longest sequence that can be stored or queried
        (shorter ones are NUL-padded; `editdistance.eval` does not need equal lengths)."""
        self._L = seq_len
        self._mode = _native.FX_LEVENSHTEIN if distance == "levenshtein" else _native.FX_HAMMING
        self._cache = _native.NativeCache(_native.Engine.get(device), seq_len)
        self._index = {}          # sequence -> position (dict semantics: re-adding updates the fitness)
        self._fitness = []
        self._fit_arr = np.empty(1024, np.float64)         # the same values as float64 (fx_cache_density), grown by doubling
        self._fit_f64 = True                              # every value so far is a float64 / Python float / int

    def _set_fit(self, pos: int, f):
        if pos >= self._fit_arr.shape[0]:
            self._fit_arr = np.concatenate([self._fit_arr, np.empty(self._fit_arr.shape[0], np.float64)])
        if type(f) in (float, np.float64, int):
            self._fit_arr[pos] = f
        else:
            self._fit_f64 = False                          # (a float32 scalar divides and adds in float32 under NumPy's rules)

    def _fit_array(self):
        """The fitness values as a float64 array; None when some value is not a float64 / Python float / int."""
        return self._fit_arr[:len(self._fitness)] if self._fit_f64 else None

    def __len__(self):
        return len(self._fitness)

    def __contains__(self, seq):
        return seq in self._index

    def __getitem__(self, seq):
        return self._fitness[self._index[seq]]

    def add(self, seq: str, fitness: float):
        """`self.all_seqs[seq] = fitness`."""
        pos = self._index.get(seq)
        if pos is not None:
            self._fitness[pos] = fitness
            self._set_fit(pos, fitness)
            return
        self._index[seq] = len(self._fitness)
        self._set_fit(len(self._fitness), fitness)
        self._fitness.append(fitness)
        self._cache.append(_native.ragged_to_bytes([seq], self._L))

    def add_many(self, seqs, fitnesses):
        """`for s, f in zip(seqs, fitnesses): self.all_seqs[s] = f` with ONE upload of the new keys (a DyNA-PPO environment step
        records its whole batch: ten appends were ten host-to-device copies, ~100 us of the step's bookkeeping)."""
        new = []
        for seq, f in zip(seqs, fitnesses):
            pos = self._index.get(seq)
            if pos is not None:
                self._fitness[pos] = f
                self._set_fit(pos, f)
                continue
            self._index[seq] = len(self._fitness)
            self._set_fit(len(self._fitness), f)
            self._fitness.append(f)
            new.append(seq)
        if new:
            self._cache.append(_native.ragged_to_bytes(new, self._L))

    @staticmethod
    def _check_radius(dist_radius):
        # the device reports min(distance, 255) in one byte (fx_cache_distances): a radius past 254 cannot be told from "far"
        if dist_radius > 254:
            raise ValueError("sequence density: dist_radius must be at most 254 (distances are kept in one byte)")

    def distances(self, seq: str) -> np.ndarray:
        return self._cache.distances(_native.ragged_to_bytes([seq], self._L), self._mode)[0]

    def densities(self, seqs, dist_radius: int = 2):
        """`[self.density(s) for s in seqs]` with ONE distance-matrix launch for the whole batch."""
        seqs = [str(s) for s in seqs]
        self._check_radius(dist_radius)
        if not seqs or len(self._fitness) == 0:
            return [0 for _ in seqs]
        # one distance launch, the radius filter and the sums in C (fx_cache_density: the Python loop's float operations in
        # insertion order); a query without neighbours keeps the int 0 the reference's `dens` starts as
        fit = self._fit_array()
        if fit is not None:
            dens, cnt = self._cache.density(_native.ragged_to_bytes(seqs, self._L), fit, dist_radius, self._mode)
            return [float(d) if n else 0 for d, n in zip(dens.tolist(), cnt.tolist())]
        # fitness values that are not float64 (a float32 scalar divides and adds in float32 under NumPy's rules): the Python
        # operations themselves, on the neighbours of all queries found with three array operations
        d_all = self._cache.distances(_native.ragged_to_bytes(seqs, self._L), self._mode)
        rows, cols = np.nonzero((d_all != 0) & (d_all <= dist_radius))
        out = [0] * len(seqs)
        vals = self._fitness
        for r, i, dist in zip(rows.tolist(), cols.tolist(), d_all[rows, cols].tolist()):
            out[r] += vals[i] / dist
        return out

    def density(self, seq: str, dist_radius: int = 2):
        """dyna_ppo.py:106-114: `dens += all_seqs[s] / dist` for 0 < dist <= radius, in insertion order."""
        dens = 0
        self._check_radius(dist_radius)
        if len(self._fitness) == 0:
            return dens
        d = self.distances(seq)
        for i in np.flatnonzero((d != 0) & (d <= dist_radius)):
            dens += self._fitness[i] / int(d[i])
        return dens
