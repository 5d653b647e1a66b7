"""float64 NumPy restatement of the reference `get_fitness` hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line (relative to /root/reference) whose behaviour it restates.
Nothing here is copied from the reference; the Keras forward is restated from
documented Keras layer semantics because TensorFlow itself is a third-party,
un-vendored dependency (setup.py:29, docs/requirements.txt:99 pins 2.3.1) that
cannot be installed in this environment -> Keras forward parity is UNPINNED.
"""
from __future__ import annotations

import numpy as np

# flexs/utils/sequence_utils.py:7-17
AAS = "ILVAGMFYWEDQNHCRKSTP"
RNAA = "UGCA"
DNAA = "TGCA"
BA = "01"


# --------------------------------------------------------------------------
# encode / decode  (flexs/utils/sequence_utils.py:32-66)
# --------------------------------------------------------------------------
def string_to_one_hot(sequence: str, alphabet: str) -> np.ndarray:
    """sequence_utils.py:32-47 -- (L, A) float64, `alphabet.index` raises
    ValueError on an unknown character."""
    out = np.zeros((len(sequence), len(alphabet)))
    for i, ch in enumerate(sequence):
        out[i, alphabet.index(ch)] = 1
    return out


def encode_batch_loop(sequences, alphabet: str) -> np.ndarray:
    """keras_model.py:70-75 -- the reference's per-character Python loop, then
    np.array -> (N, L, A) float64 -> float32 (tf.convert_to_tensor dtype)."""
    return np.array([string_to_one_hot(s, alphabet) for s in sequences]).astype(
        np.float32
    )


def encode_codes(sequences, alphabet: str) -> np.ndarray:
    """Vectorised char -> alphabet index, (N, L) uint8.  Same ValueError rule."""
    seqs = [str(s) for s in sequences]
    if len(seqs) == 0:
        return np.zeros((0, 0), np.uint8)
    L = len(seqs[0])
    if any(len(s) != L for s in seqs):
        raise ValueError("ragged sequence batch")
    raw = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8).reshape(-1, L)
    lut = np.full(256, 255, np.uint8)
    # alphabet.index returns the FIRST occurrence -> fill in reverse order
    for i in range(len(alphabet) - 1, -1, -1):
        lut[ord(alphabet[i])] = i
    codes = lut[raw]
    if (codes == 255).any():
        raise ValueError("substring not found")
    return codes


def encode_batch(sequences, alphabet: str) -> np.ndarray:
    """Vectorised equivalent of encode_batch_loop, float64 (N, L, A)."""
    codes = encode_codes(sequences, alphabet)
    return np.eye(len(alphabet))[codes]


def one_hot_to_string(one_hot, alphabet: str) -> str:
    """sequence_utils.py:50-66 -- per-position argmax (first max wins)."""
    idx = np.argmax(one_hot, axis=1)
    return "".join(alphabet[i] for i in idx)


# --------------------------------------------------------------------------
# weight containers (Keras `get_weights()` order)
# --------------------------------------------------------------------------
def cnn_shapes(L, A, F, H, K):
    """cnn.py:23-54"""
    return [
        (K, A, F), (F,),          # Conv1D valid
        (K, F, F), (F,),          # Conv1D same
        (A - 1, F, F), (F,),      # Conv1D same, kernel = len(alphabet) - 1
        (F, H), (H,),
        (H, H), (H,),
        (H, 1), (1,),
    ]


def mlp_shapes(L, A, H):
    """mlp.py:21-31"""
    return [(L * A, H), (H,), (H, H), (H,), (H, H), (H,), (H, 1), (1,)]


def ge_shapes(L, A, H):
    """global_epistasis_model.py:26-36"""
    return [(L * A, 1), (1,), (1, H), (H,), (H, H), (H,), (H, 1), (1,)]


def synth_weights(shapes, seed: int, bias_scale: float = 0.1):
    """Glorot-uniform kernels, U(-0.1, 0.1) biases
    (non-zero so every bias path is exercised), default_rng(seed)."""
    rng = np.random.default_rng(seed)
    out = []
    for shp in shapes:
        if len(shp) == 1:
            out.append(rng.uniform(-bias_scale, bias_scale, shp).astype(np.float32))
        else:
            receptive = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
            fan_in, fan_out = shp[-2] * receptive, shp[-1] * receptive
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out.append(rng.uniform(-lim, lim, shp).astype(np.float32))
    return out


# --------------------------------------------------------------------------
# Keras layer semantics in float64
# --------------------------------------------------------------------------
def _relu(x):
    return np.maximum(x, 0.0)


def conv1d(x, w, b, padding: str):
    """Keras Conv1D(strides=1): cross-correlation, kernel (k, Cin, Cout).
    valid: L_out = L - k + 1 (ValueError if L < k).
    same:  L_out = L, pad_left = (k-1)//2, pad_right = k-1-pad_left."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    b = np.asarray(b, np.float64)
    k = w.shape[0]
    N, L, _ = x.shape
    if padding == "valid":
        if L < k:
            raise ValueError("Negative dimension size: valid conv with L < kernel_size")
        Lout = L - k + 1
        xp = x
    elif padding == "same":
        pl = (k - 1) // 2
        pr = k - 1 - pl
        xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
        Lout = L
    else:
        raise ValueError(padding)
    out = np.broadcast_to(b, (N, Lout, w.shape[2])).copy()
    for j in range(k):
        out += xp[:, j:j + Lout, :] @ w[j]
    return out


def cnn_forward(x, weights):
    """cnn.py:23-54 at predict time.  x (N, L, A) -> (N,) float64."""
    w1, b1, w2, b2, w3, b3, d1, c1, d2, c2, d3, c3 = [np.asarray(a, np.float64) for a in weights]
    h = _relu(conv1d(x, w1, b1, "valid"))
    h = _relu(conv1d(h, w2, b2, "same"))
    # MaxPooling1D(1): pool 1 / stride 1 -> identity (cnn.py:40)
    h = _relu(conv1d(h, w3, b3, "same"))
    h = h.max(axis=1)                      # GlobalMaxPooling1D (cnn.py:48)
    h = _relu(h @ d1 + c1)
    h = _relu(h @ d2 + c2)
    # Dropout(0.25) inactive in predict (cnn.py:51)
    return (h @ d3 + c3)[:, 0]


def mlp_forward(x, weights):
    """mlp.py:21-31.  Flatten is row-major: feature l*A + a."""
    d1, c1, d2, c2, d3, c3, d4, c4 = [np.asarray(a, np.float64) for a in weights]
    h = np.asarray(x, np.float64).reshape(x.shape[0], -1)
    h = _relu(h @ d1 + c1)
    h = _relu(h @ d2 + c2)
    h = _relu(h @ d3 + c3)
    return (h @ d4 + c4)[:, 0]


def ge_forward(x, weights):
    """global_epistasis_model.py:26-36."""
    return mlp_forward(x, weights)          # same stack; first Dense has 1 unit


FORWARD = {"cnn": cnn_forward, "mlp": mlp_forward, "ge": ge_forward}


def nan_to_num_f32(y):
    """keras_model.py:77-78: predict -> (N,1) float32 -> squeeze -> nan_to_num."""
    return np.nan_to_num(np.asarray(y).astype(np.float32))


def keras_fitness(sequences, alphabet, kind, weights, exact=False):
    """keras_model.py:69-79 restated.  Returns float32 (N,) like the reference;
    with exact=True returns the float64 value before the float32 cast (used to
    measure the kernel's error against an exactly-rounded target)."""
    x = encode_batch(sequences, alphabet)
    y = FORWARD[kind](x, weights)
    return y if exact else nan_to_num_f32(y)


# --------------------------------------------------------------------------
# Ensemble reductions
# --------------------------------------------------------------------------
def ensemble_mean(scores_NM):
    """ensemble.py:24,59 -- default combine_with = np.mean(x, axis=1) on the
    stacked (N, M) array; the oracle uses NumPy itself so summation order
    (pairwise for M >= 8) is the reference's by construction."""
    return np.mean(scores_NM, axis=1)


def np_pairwise_sum_f32(row):
    """Explicit restatement of NumPy's pairwise summation for a contiguous
    float32 vector (numpy/_core/src/umath/loops_utils.h.src `pairwise_sum`,
    numpy 2.2): n < 8 sequential; n <= 128 eight interleaved accumulators +
    tail; else recursive halves.  Used to document / test the order the device
    reduce kernel reproduces bit-exactly."""
    row = np.asarray(row, np.float32)
    n = row.shape[0]
    f = np.float32
    if n < 8:
        # numpy's reduce loop seeds the accumulator with row[0] and adds the
        # rest; pairwise_sum itself starts from 0.0 -- identical results except
        # for -0.0, which never appears here.
        res = f(0.0)
        for i in range(n):
            res = f(res + row[i])
        return res
    if n <= 128:
        r = [row[i] for i in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = f(r[k] + row[i + k])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + row[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(np_pairwise_sum_f32(row[:n2]) + np_pairwise_sum_f32(row[n2:]))


def r2_weights(model_preds, labels):
    """adaptive_ensemble.py:12-26."""
    import scipy.stats

    r2s = np.array([scipy.stats.pearsonr(p, labels)[0] ** 2 for p in model_preds])
    return r2s / r2s.sum()


def adaptive_combine(weights, scores_NM):
    """adaptive_ensemble.py:54,102 -- np.sum(w * x, axis=1)."""
    return np.sum(weights * scores_NM, axis=1)


# --------------------------------------------------------------------------
# edit distance + NoisyAbstractModel
# --------------------------------------------------------------------------
def levenshtein(a: str, b: str) -> int:
    """Unit-cost Levenshtein distance == `editdistance.eval` (third-party,
    setup.py:23; docs/requirements.txt:23 pins 0.5.3), textbook two-row DP."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def hamming(a: str, b: str) -> int:
    if len(a) != len(b):
        raise ValueError("hamming needs equal lengths")
    return sum(x != y for x, y in zip(a, b))


def min_distance(sequence, cache_keys, dist=levenshtein):
    """noisy_abstract_model.py:42-60: first cache entry (insertion order) that
    attains the minimum; early exit on distance 1; empty cache -> (0, seq)."""
    if len(cache_keys) == 0:
        return 0, sequence
    best, closest = np.inf, None
    for s in cache_keys:
        d = dist(sequence, s)
        if d == 1:
            return d, s
        if d < best:
            best, closest = d, s
    return best, closest


class NoisyAbstractModelOracle:
    """noisy_abstract_model.py:9-101 restated (uses the global legacy NumPy RNG
    exactly like the reference: one np.random.exponential per uncached query,
    in batch order; np.random.choice over cache values when the neighbour
    fitness is negative)."""

    def __init__(self, landscape, signal_strength=0.9, dist=levenshtein):
        self.name = f"NAMb_ss{signal_strength}"
        self.cost = 0
        self.landscape = landscape
        self.ss = signal_strength
        self.cache = {}
        self._dist = dist

    def train(self, sequences, labels):
        self.cache.update(zip(sequences, labels))        # :62-67

    def get_fitness(self, sequences):
        self.cost += len(sequences)                       # landscape.py:44
        return self._fitness_function(sequences)

    def _fitness_function(self, sequences):
        sequences = np.array(sequences)
        fitnesses = np.empty(len(sequences))
        cached = np.array([s in self.cache for s in sequences], dtype=bool)
        fitnesses[cached] = np.array([self.cache[s] for s in sequences[cached]])
        new = []
        for seq in sequences[~cached]:
            d, nb = min_distance(seq, list(self.cache), self._dist)
            signal = self.landscape.get_fitness([seq]).item()
            nbf = self.landscape.get_fitness([nb]).item()
            if nbf >= 0:
                noise = np.random.exponential(scale=nbf)
            else:
                noise = np.random.choice(list(self.cache.values()))
            alpha = self.ss ** d
            new.append(alpha * signal + (1 - alpha) * noise)
        fitnesses[~cached] = new
        self.cache.update(zip(sequences[~cached], fitnesses[~cached]))
        return np.array(fitnesses)


# --------------------------------------------------------------------------
# additive landscape
# --------------------------------------------------------------------------
class AdditiveAAVOracle:
    """flexs/landscapes/additive_aav_packaging.py:25-119 restated: `data` is the parsed single-substitution
    file {position(str): {residue: {"log2_<phenotype>_v_wt": x, "log2_packaging_v_wt": y}}}.
    Pinned by tests/golden/additive_aav.json (outputs of the reference class on a synthetic data file)."""

    def __init__(self, data, phenotype="heart", minimum_fitness_multiplier=1, start=0, end=735, noise=0):
        self.name = f"AdditiveAAVPackaging_phenotype={phenotype}"
        self.cost = 0
        self.key = f"log2_{phenotype}_v_wt"
        self.mfm, self.start, self.end, self.noise = minimum_fitness_multiplier, start, end, noise
        self.data = {int(p): v for p, v in data.items() if start <= int(p) < end}          # :66-76
        seq, total = "", 0
        for pos in self.data:                                                              # :80-99
            top, top_aa = -10, "M"
            for aa in self.data[pos]:
                fit = self.data[pos][aa][self.key]
                if fit > top and self.data[pos][aa]["log2_packaging_v_wt"] > -6:
                    top, top_aa = fit, aa
            seq += top_aa
            total += top
        self.top_seq, self.max_possible = seq, total

    def raw(self, seq):
        total = 0                                                                          # :101-107
        for i, s in enumerate(seq):
            if s in self.data[self.start + i]:
                total += self.data[self.start + i][s][self.key]
        return total + self.mfm * self.max_possible

    def get_fitness(self, sequences):
        self.cost += len(sequences)
        out = []
        for seq in sequences:                                                              # :109-119
            normed = self.raw(seq) / (self.max_possible * (self.mfm + 1))
            out.append(max(0, normed + np.random.normal(scale=self.noise)))
        return np.array(out)
