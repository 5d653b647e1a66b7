"""Synthetic workloads for benchmarks and smoke runs:
random sequences over an alphabet, Glorot-uniform kernels with non-zero biases
so that every bias path is exercised.  No dataset / checkpoint access needed."""
import numpy as np


def random_sequence_bytes(n: int, L: int, alphabet: str, seed: int) -> np.ndarray:
    """(n, L) uint8 ASCII codes of uniformly random sequences, default_rng(seed)."""
    codes = np.random.default_rng(seed).integers(0, len(alphabet), (n, L))
    return np.frombuffer(alphabet.encode("latin-1"), np.uint8)[codes]


def bytes_to_strings(seq_bytes: np.ndarray):
    L = seq_bytes.shape[1]
    flat = np.ascontiguousarray(seq_bytes).tobytes().decode("latin-1")
    return [flat[i * L:(i + 1) * L] for i in range(seq_bytes.shape[0])]


def synthetic_weights(shapes, seed: int, bias_scale: float = 0.1):
    rng = np.random.default_rng(seed)
    out = []
    for shp in shapes:
        if len(shp) == 1:
            out.append(rng.uniform(-bias_scale, bias_scale, shp).astype(np.float32))
        else:
            receptive = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
            lim = np.sqrt(6.0 / (shp[-2] * receptive + shp[-1] * receptive))
            out.append(rng.uniform(-lim, lim, shp).astype(np.float32))
    return out


def algorithmic_macs(kind: str, L: int, A: int, H: int, F: int = 0, K: int = 0) -> int:
    """Dense multiply-accumulates per sequence per member,
    NOT discounted for one-hot sparsity or zero padding."""
    if kind == "cnn":
        L1 = L - K + 1
        return L1 * K * A * F + L1 * K * F * F + L1 * (A - 1) * F * F + F * H + H * H + H
    if kind == "mlp":
        return L * A * H + 2 * H * H + H
    return L * A + H + H * H + H
