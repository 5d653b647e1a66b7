"""Round-2 soak: the seeded random-shape sweep of tests/test_gpu_parity.py over many more seeds, plus a second generator biased
towards the shapes the small-launch forms serve (4-letter CNN of every length with K = 5 / H = 100, protein CNN, 1-3 tiles)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_parity as T
from flexs_amd import _native
from flexs_amd.utils import sequence_utils as s_utils
from oracle import ref_np

eng = _native.Engine.get(0)
eng.set_option("poison_outputs", 1)
fails = 0


def check(kind, alpha, A, L, H, F, K, M, n, seed, tag):
    global fails
    try:
        natives, ws = zip(*[T.make_native(eng, kind, L, A, H, F, K, seed=500 + 7 * seed + m) for m in range(M)])
        lut = _native.make_lut(alpha)
        b, seqs = T.rand_seqs(n, L, alpha, seed=seed)
        got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
        for m in range(M):
            T.assert_scores(got[:, m], ref_np.keras_fitness(seqs, alpha, kind, ws[m], exact=True), f"{tag} member {m}")
        assert np.array_equal(mean, np.mean(got, axis=1))
        rng = np.random.default_rng(seed)
        for k in sorted({1, min(n, 20), int(rng.integers(1, n + 1))}):
            part, _ = eng.score(list(natives), b[:k], lut)
            assert np.array_equal(part, got[:k]), f"prefix {k}"
    except Exception as ex:                                   # noqa: BLE001
        fails += 1
        print("FAIL", tag, kind, repr(alpha), L, H, F, K, M, n, "->", str(ex)[:200], flush=True)


for seed in range(64, 364):
    kind, alpha, A, L, H, F, K, M, n = T._random_case(seed)
    check(kind, alpha, A, L, H, F, K, M, n, seed, f"sweep seed {seed}")
rng = np.random.default_rng(7)
for i in range(300):
    if rng.random() < 0.7:
        alpha, L = ("TGCA", "UGCA")[int(rng.integers(0, 2))], int(rng.integers(5, 130))
    else:
        alpha, L = s_utils.AAS, int(rng.integers(5, 100))
    M, n = int(rng.integers(1, 5)), int(rng.choice([1, 3, 16, 17, 20, 40, 48, 100, 400, 1400]))
    check("cnn", alpha, len(alpha), L, 100, 32, 5, M, n, 10_000 + i, f"small-launch case {i}")
print("soak done, failures:", fails, flush=True)
