"""Parity tests proper: the HIP path (through the C ABI) against the oracle, the
committed golden fixtures, and size-independent properties at BASELINE sizes.

Tolerance for fp32 fitness scores (BASELINE.json north_star: "within 1e-5
relative"): |gpu - oracle_f64| <= 1e-5 * |oracle| + 2.5e-7  element-wise
(np.allclose form; the absolute term covers scores that cancel to ~0).
Integer / float64 paths (distances, ensemble mean, NAM blend, codecs) are
bit-exact.
"""
import json
import os

import numpy as np
import pytest

import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
from oracle import c_oracle, ref_np

from gpu_common import ATOL, ERROR_STATS, RTOL, ab_option, assert_scores, close, eng, make_native, rand_seqs  # noqa: F401  (eng: the session fixture)

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ MFMA layout ground truth
def test_mfma_operand_layout(eng):
    """The lane layout of v_mfma_f32_16x16x4_f32 assumed by the kernels and by
    tests/mfma_sim.py, checked on the hardware with asymmetric operands."""
    import mfma_sim

    rng = np.random.default_rng(0)
    a = rng.integers(-4, 5, 64).astype(np.float32)
    b = rng.integers(-4, 5, 64).astype(np.float32)
    c = rng.integers(-4, 5, (64, 4)).astype(np.float32)
    got = eng.mfma_probe(a, b, c)
    want = mfma_sim.mfma16(a.astype(np.float64), b.astype(np.float64), c.astype(np.float64))
    assert np.array_equal(got, want.astype(np.float32))


def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()


# ------------------------------------------------------------------ CNN
@pytest.mark.parametrize("conv1_mfma", [0, 1])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 7, 11])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 33, 1000, 4099])
def test_cnn_l8_variants_and_tails(eng, variant, n, conv1_mfma):
    """BASELINE configs[0]/[1] shape: TF-binding L=8, alphabet TGCA, CNN(32,100,k5);
    every launch geometry (variant) x both forms of the one-hot conv1 (LDS gather / MFMA)."""
    eng.set_option("force_generic", 0)
    if not (ab_option(eng, "cnn_variant", variant) and ab_option(eng, "cnn_conv1_mfma", conv1_mfma)):
        eng.set_option("cnn_variant", 0)
        pytest.skip("a kernel form of the A/B build (make ab)")
    nm, w = make_native(eng, "cnn", 8, 4, 100, 32, 5)
    b, seqs = rand_seqs(n, 8, "TGCA", seed=n)
    got, _ = eng.score([nm], b, _native.make_lut("TGCA"))
    want = ref_np.keras_fitness(seqs, "TGCA", "cnn", w, exact=True)
    assert_scores(got[:, 0], want, f"cnn L8 variant {variant} n={n}")
    eng.set_option("cnn_variant", 0)
    eng.set_option("cnn_conv1_mfma", 0)


@pytest.mark.parametrize("n", [1, 31, 5000, 70000])
def test_cnn_l14_unrolled_specialisation(eng, n):
    """RNA L=14: variant 6 (unrolled position loop) vs the oracle and vs the dynamic-loop kernel."""
    nm, w = make_native(eng, "cnn", 14, 4, 100, 32, 5, seed=21)
    b, seqs = rand_seqs(n, 14, "UGCA", seed=n)
    lut = _native.make_lut("UGCA")
    eng.set_option("cnn_variant", 10)
    got10, _ = eng.score([nm], b, lut)
    if ab_option(eng, "cnn_variant", 6):                 # (the unrolled form without s_setprio: A/B build)
        got6, _ = eng.score([nm], b, lut)
        assert np.array_equal(got6, got10)               # s_setprio changes scheduling only
    eng.set_option("cnn_variant", 4)
    got4, _ = eng.score([nm], b, lut)
    eng.set_option("cnn_variant", 0)
    want = ref_np.keras_fitness(seqs, "UGCA", "cnn", w, exact=True)
    assert_scores(got10[:, 0], want, f"cnn L14 variant 10 n={n}")
    assert_scores(got4[:, 0], want, f"cnn L14 variant 4 n={n}")


@pytest.mark.parametrize("L,A,alpha,n", [(8, 4, "TGCA", 3000), (5, 4, "TGCA", 500), (6, 4, "TGCA", 500),
                                         (14, 4, "UGCA", 2000), (50, 4, "UGCA", 600), (100, 4, "UGCA", 300),
                                         (20, 20, s_utils.AAS, 400), (66, 20, s_utils.AAS, 200),
                                         (90, 20, s_utils.AAS, 150), (5, 20, s_utils.AAS, 1000), (23, 20, s_utils.AAS, 4097)])
def test_cnn_mfma_vs_oracle(eng, L, A, alpha, n):
    eng.set_option("force_generic", 0)
    nm, w = make_native(eng, "cnn", L, A, 100, 32, 5, seed=7)
    b, seqs = rand_seqs(n, L, alpha, seed=L)
    got, _ = eng.score([nm], b, _native.make_lut(alpha))
    want = ref_np.keras_fitness(seqs, alpha, "cnn", w, exact=True)
    assert_scores(got[:, 0], want, f"cnn mfma L={L} A={A}")
    if A == 20 and ab_option(eng, "cnn_pair", 0):        # single-wave-per-tile form of the wide-alphabet kernel (A/B build)
        got_s, _ = eng.score([nm], b, _native.make_lut(alpha))
        eng.set_option("cnn_pair", 1)
        assert_scores(got_s[:, 0], want, f"cnn single-wave form L={L} A={A}")
    if ab_option(eng, "cnn_conv1_mfma", 1):              # one-hot conv1 on the MFMA pipe instead of the LDS gather (A/B build)
        eng.set_option("cnn_pair", 0)
        got_m, _ = eng.score([nm], b, _native.make_lut(alpha))
        eng.set_option("cnn_conv1_mfma", 0)
        eng.set_option("cnn_pair", 1)
        assert_scores(got_m[:, 0], want, f"cnn mfma(conv1 on mfma) L={L} A={A}")
    # the shape-agnostic kernel must agree too (independent on-device implementation)
    eng.set_option("force_generic", 1)
    got_g, _ = eng.score([nm], b, _native.make_lut(alpha))
    eng.set_option("force_generic", 0)
    assert_scores(got_g[:, 0], want, f"cnn generic L={L} A={A}")


@pytest.mark.parametrize("kind", ["cnn", "mlp", "ge"])
@pytest.mark.parametrize("H", [1, 16, 20, 50, 64, 113, 128, 200, 208, 209, 256, 257, 300])
def test_hidden_sizes(eng, kind, H):
    """Hidden sizes are rounded up to an instantiated tile count (1, 2, 4, 7, 8, 13 x 16) on the MFMA
    path (H <= 256; HxH blocks stream from L2 beyond 128); larger ones use the shape-agnostic kernels."""
    for L, A, alpha in ((8, 4, "TGCA"), (12, 20, s_utils.AAS)):
        nm, w = make_native(eng, kind, L, A, H, 32 if kind == "cnn" else 0, 5 if kind == "cnn" else 0, seed=H)
        b, seqs = rand_seqs(700, L, alpha, seed=H + L)
        got, _ = eng.score([nm], b, _native.make_lut(alpha))
        want = ref_np.keras_fitness(seqs, alpha, kind, w, exact=True)
        assert_scores(got[:, 0], want, f"{kind} H={H} L={L} A={A}")


@pytest.mark.parametrize("L", [237, 238])
def test_cnn_gfp_length(eng, L):
    """BASELINE configs[4] shape (GFP: 238 residues in the reference, 237 in BASELINE.json), reduced N."""
    nm, w = make_native(eng, "cnn", L, 20, 100, 32, 5, seed=3)
    b, seqs = rand_seqs(96, L, s_utils.AAS, seed=L)
    got, _ = eng.score([nm], b, _native.make_lut(s_utils.AAS))
    want = ref_np.keras_fitness(seqs, s_utils.AAS, "cnn", w, exact=True)
    assert_scores(got[:, 0], want, f"cnn L={L}")


@pytest.mark.parametrize("L,A,alpha,F,H,K", [(3, 4, "TGCA", 1, 1, 2), (9, 4, "TGCA", 8, 20, 4), (12, 2, "01", 16, 30, 3),
                                             (10, 4, "TGCA", 32, 100, 3), (8, 4, "TGCA", 32, 200, 5),
                                             (5, 4, "TGCA", 32, 100, 5), (7, 20, s_utils.AAS, 4, 130, 2)])
def test_cnn_odd_shapes_generic(eng, L, A, alpha, F, H, K):
    """Shapes outside the MFMA instantiations (incl. the reference's own smoke test
    CNN(seq_len=3, num_filters=1, hidden_size=1, kernel_size=2) and even kernels)."""
    nm, w = make_native(eng, "cnn", L, A, H, F, K, seed=5)
    b, seqs = rand_seqs(257, L, alpha, seed=1)
    got, _ = eng.score([nm], b, _native.make_lut(alpha))
    want = ref_np.keras_fitness(seqs, alpha, "cnn", w, exact=True)
    assert_scores(got[:, 0], want, f"cnn odd {L},{A},{F},{H},{K}")


# ------------------------------------------------------------------ MLP / GE
@pytest.mark.parametrize("kind", ["mlp", "ge"])
@pytest.mark.parametrize("L,A,alpha,H,n", [(14, 4, "UGCA", 100, 3000), (8, 4, "TGCA", 100, 1000), (14, 4, "UGCA", 97, 300),
                                           (90, 20, s_utils.AAS, 100, 1000), (237, 20, s_utils.AAS, 100, 200),
                                           (14, 4, "UGCA", 200, 300), (3, 4, "TGCA", 1, 50), (10, 2, "01", 100, 100),
                                           (14, 4, "UGCA", 104, 200), (14, 4, "UGCA", 109, 200), (50, 4, "UGCA", 100, 500),
                                           (11, 3, "ABC", 100, 300)])
def test_mlp_ge_vs_oracle(eng, kind, L, A, alpha, H, n):
    for force, l1 in ((0, 0), (0, 1), (1, 0)):
        eng.set_option("force_generic", force)
        if not ab_option(eng, "mlp_l1_mfma", l1):
            continue                                     # (the MFMA form of the one-hot first layer: A/B build)
        nm, w = make_native(eng, kind, L, A, H, seed=11)
        b, seqs = rand_seqs(n, L, alpha, seed=L + H)
        got, _ = eng.score([nm], b, _native.make_lut(alpha))
        want = ref_np.keras_fitness(seqs, alpha, kind, w, exact=True)
        assert_scores(got[:, 0], want, f"{kind} L={L} A={A} H={H} generic={force} l1_mfma={l1}")
    eng.set_option("force_generic", 0)
    eng.set_option("mlp_l1_mfma", 0)


@pytest.mark.parametrize("M", [1, 2, 3, 8, 11, 17])
def test_ensemble_matrix_and_numpy_order_mean(eng, M):
    L, alpha = 8, "TGCA"
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, 100, 32, 5, seed=1000 + m) for m in range(M)])
    b, seqs = rand_seqs(2049, L, alpha, seed=M)
    nm, mean = eng.score(list(natives), b, _native.make_lut(alpha), want_matrix=True, want_mean=True)
    assert nm.shape == (2049, M)
    for m in range(M):
        assert_scores(nm[:, m], ref_np.keras_fitness(seqs, alpha, "cnn", ws[m], exact=True), f"member {m}")
    assert np.array_equal(mean, np.mean(nm, axis=1)), "device mean must be np.mean bit-for-bit (ensemble.py:24)"
    assert np.array_equal(eng.ensemble_mean(nm), np.mean(nm, axis=1))


def test_reduce_kernel_golden(eng, golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "ensemble.json")))
    arrs = np.load(os.path.join(golden_dir, "ensemble.npz"))
    for ci, case in enumerate(meta["cases"]):
        if case["dtype"] != "float32":
            continue
        assert np.array_equal(eng.ensemble_mean(arrs[f"in{ci}"]), arrs[f"out{ci}"]), case
    got = eng.ensemble_weighted_sum(arrs["ada_in"], arrs["ada_w"])
    assert got.dtype == np.float64 and np.array_equal(got, arrs["ada_out_w"])
    assert np.array_equal(eng.ensemble_weighted_sum(arrs["ada_in"], np.ones(4) / 4), arrs["ada_out_default"])
    rng = np.random.default_rng(0)
    for M in (7, 8, 9, 16, 100, 129, 300):
        x = (rng.standard_normal((513, M)) * rng.choice([1e-3, 1, 1e3], (513, M))).astype(np.float32)
        assert np.array_equal(eng.ensemble_mean(x), np.mean(x, axis=1)), M
        w = rng.random(M)
        assert np.array_equal(eng.ensemble_weighted_sum(x, w), np.sum(w * x, axis=1)), M


# ------------------------------------------------------------------ codecs
def test_encode_decode_golden(eng, golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "encode.json")))
    for case in fx["cases"]:
        alpha = fx["alphabets"][case["alphabet"]] if "alphabet" in case else case["alphabet_literal"]
        got = s_utils.string_to_one_hot(case["sequence"], alpha)
        assert got.dtype == np.float64 and np.array_equal(got, np.array(case["one_hot"], dtype=np.float64))
    meta = json.load(open(os.path.join(golden_dir, "decode.json")))
    arrs = np.load(os.path.join(golden_dir, "decode.npz"))
    alphabets = {"AAS": s_utils.AAS, "RNAA": s_utils.RNAA, "DNAA": s_utils.DNAA}
    for i, (an, want) in enumerate(zip(meta["alphabet"], meta["strings"])):
        assert s_utils.one_hot_to_string(arrs[f"x{i}"], alphabets[an]) == want
    rng = np.random.default_rng(0)
    for L, A, alpha in ((8, 4, "TGCA"), (90, 20, s_utils.AAS), (7, 2, "01")):
        b, seqs = rand_seqs(1001, L, alpha, seed=L)
        oh = s_utils.strings_to_one_hot(seqs, alpha)
        assert oh.dtype == np.float32 and np.array_equal(oh, ref_np.encode_batch(seqs, alpha).astype(np.float32))
        assert s_utils.one_hots_to_strings(oh, alpha) == seqs                  # round trip
        x = rng.standard_normal((40, L, A))
        x[::3] = np.round(x[::3])
        x[5, 2, 1] = np.nan
        assert s_utils.one_hots_to_strings(x, alpha) == [ref_np.one_hot_to_string(r, alpha) for r in x]


# ------------------------------------------------------------------ BASELINE sizes: properties
def test_baseline_config1_full_batch(eng):
    """configs[1]: TF-binding L=8, 3-member CNN ensemble, batch = 1e5 on one GPU.
    Full oracle comparison (the float64 NumPy oracle does 3e5 forwards in seconds) plus
    size-independent properties: permutation equivariance, duplicates, chunk invariance."""
    L, alpha, N, M = 8, "TGCA", 100_000, 3
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, 100, 32, 5, seed=1000 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(N, L, alpha, seed=0)
    nm, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    x = ref_np.encode_batch(seqs, alpha)
    for m in range(M):
        assert_scores(nm[:, m], ref_np.cnn_forward(x, ws[m]), f"full batch member {m}")
    assert np.array_equal(mean, np.mean(nm, axis=1))
    perm = np.random.default_rng(1).permutation(N)
    nm_p, mean_p = eng.score(list(natives), b[perm], lut, want_matrix=True, want_mean=True)
    assert np.array_equal(nm_p, nm[perm]) and np.array_equal(mean_p, mean[perm])      # row independence, bit-exact
    nm_c = np.concatenate([eng.score(list(natives), b[i:i + 33_333], lut)[0] for i in range(0, N, 33_333)])
    assert np.array_equal(nm_c, nm)                                                    # chunk invariance
    # all 4^8 8-mers exist in TF-binding: identical sequences must score identically
    _, inv = np.unique(b, axis=0, return_inverse=True)
    order = np.argsort(inv, kind="stable")
    same = inv[order][1:] == inv[order][:-1]
    assert np.array_equal(mean[order][1:][same], mean[order][:-1][same])


def test_baseline_config3_and_4_shapes(eng):
    """configs[2] (RNA L=14: MLP + NoisyAbstractModel) and configs[3] (AAV L=90, A=20:
    8-member GlobalEpistasis ensemble) at one-GPU batch sizes."""
    b, seqs = rand_seqs(100_000, 14, "UGCA", seed=3)
    nm, w = make_native(eng, "mlp", 14, 4, 100, seed=1)
    got, _ = eng.score([nm], b, _native.make_lut("UGCA"))
    assert_scores(got[:, 0], ref_np.keras_fitness(seqs, "UGCA", "mlp", w, exact=True), "C3 mlp")
    b, seqs = rand_seqs(100_000, 90, s_utils.AAS, seed=4)
    natives, ws = zip(*[make_native(eng, "ge", 90, 20, 100, seed=2000 + m) for m in range(8)])
    nm8, mean = eng.score(list(natives), b, _native.make_lut(s_utils.AAS), want_matrix=True, want_mean=True)
    codes = ref_np.encode_codes(seqs, s_utils.AAS)
    for m in range(8):
        assert_scores(nm8[:, m], c_oracle.forward("ge", codes, 20, ws[m]), f"C4 ge member {m}")
    assert np.array_equal(mean, np.mean(nm8, axis=1))


def test_baseline_config4_ten_million(eng):
    """configs[3] at its upper size (N in {1e5, 1e7}): 1e7 AAV-length sequences x 8
    GlobalEpistasis members through the host entry point (0.9 GB in, 0.36 GB out).  Oracle on a sample,
    chunk invariance against separately scored slices, equal rows -> equal scores."""
    L, alpha, N, M = 90, s_utils.AAS, 10_000_000, 8
    natives, ws = zip(*[make_native(eng, "ge", L, 20, 100, seed=2000 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    rng = np.random.default_rng(12)
    b = np.frombuffer(alpha.encode(), np.uint8)[rng.integers(0, 20, (N, L), dtype=np.uint8)]
    b[N - 1000:] = b[:1000]
    nm, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    assert nm.shape == (N, M) and np.isfinite(nm).all()
    assert np.array_equal(nm[N - 1000:], nm[:1000])
    sample = rng.choice(N, 4000, replace=False)
    codes = lut[b[sample]]
    for m in range(M):
        assert_scores(nm[sample, m], c_oracle.forward("ge", codes, 20, ws[m]), f"C4 1e7 member {m}")
    assert np.array_equal(mean[sample], np.mean(nm[sample], axis=1))
    for lo in (0, 3_333_333, N - 70_001):
        part, _ = eng.score(list(natives), b[lo:lo + 70_001], lut)
        assert np.array_equal(part, nm[lo:lo + 70_001])


def test_baseline_config5_one_gpu_share(eng):
    """configs[4]: GFP L=237, A=20, 3-member CNN ensemble; one GPU's share of the 5e5 virtual screen
    (62 500 sequences, pair kernel).  Oracle on a 600-row sample (6.5 MMAC per sequence and member),
    size-independent properties on the whole batch: row independence under permutation, chunk
    invariance, equal rows -> equal scores, mean == NumPy's, all bit-exact."""
    L, alpha, N, M = 237, s_utils.AAS, 62_500, 3
    natives, ws = zip(*[make_native(eng, "cnn", L, 20, 100, 32, 5, seed=3000 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, _ = rand_seqs(N, L, alpha, seed=9)
    b[1000:1200] = b[:200]                                              # planted duplicates
    nm, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    assert np.isfinite(nm).all() and np.array_equal(mean, np.mean(nm, axis=1))
    assert np.array_equal(nm[1000:1200], nm[:200])
    sample = np.random.default_rng(2).choice(N, 600, replace=False)
    codes = lut[b[sample]]
    for m in range(M):
        assert_scores(nm[sample, m], c_oracle.forward("cnn", codes, 20, ws[m]), f"C5 member {m}")
    perm = np.random.default_rng(3).permutation(N)
    nm_p, _ = eng.score(list(natives), b[perm], lut)
    assert np.array_equal(nm_p, nm[perm])
    nm_c = np.concatenate([eng.score(list(natives), b[i:i + 20_011], lut)[0] for i in range(0, N, 20_011)])
    assert np.array_equal(nm_c, nm)


def test_baseline_config5_full_virtual_screen(eng):
    """configs[4] at its FULL size on one GPU: 5e5 GFP-length sequences x the 3-member CNN ensemble (118 MB of sequence
    bytes in, 19.5 PFLOP... 0.14 s of kernel), through the public `Ensemble.get_fitness(list[str])`: properties on every
    row (duplicates -> equal scores, mean == np.mean of the stacked matrix, finite), the oracle on a sample, and the
    rows a one-GPU share holds (62 500) reproduce bit for bit inside the big batch."""
    L, alpha, N, M = 237, s_utils.AAS, 500_000, 3
    members = [bm.CNN(L, 32, 100, alpha, seed=3000 + m) for m in range(M)]
    ens = flexs_amd.Ensemble(members)
    b, _ = rand_seqs(N, L, alpha, seed=19)
    b[400_000:400_300] = b[:300]
    seqs = synth.bytes_to_strings(b)
    mean = ens.get_fitness(seqs)
    assert mean.shape == (N,) and mean.dtype == np.float32 and np.isfinite(mean).all()
    assert np.array_equal(mean[400_000:400_300], mean[:300])
    assert ens.cost == N and all(m.cost == N for m in members)
    natives = [m.native() for m in members]
    nm, mean_dev = eng.score(natives, b, members[0]._lut, want_matrix=True, want_mean=True)
    assert np.array_equal(mean_dev, mean) and np.array_equal(np.mean(nm, axis=1), mean)
    share, _ = eng.score(natives, b[187_500:250_000], members[0]._lut)          # rank 3's rows of an 8-GPU split
    assert np.array_equal(share, nm[187_500:250_000])
    sample = np.random.default_rng(6).choice(N, 300, replace=False)
    codes = members[0]._lut[b[sample]]
    for m in range(M):
        assert_scores(nm[sample, m], c_oracle.forward("cnn", codes, 20, members[m].model.get_weights()), f"C5 full member {m}")


def test_baseline_config5_eight_members_and_gfp_238(eng):
    """configs[4]'s 8-member variant at one GPU's share (62 500 x 8) and the reference's own GFP length (238 residues,
    SURVEY 8d) at 1e4 sequences: oracle on samples, row independence, mean == np.mean."""
    for L, N, M in ((237, 62_500, 8), (238, 10_000, 3)):
        alpha = s_utils.AAS
        natives, ws = zip(*[make_native(eng, "cnn", L, 20, 100, 32, 5, seed=4000 + m) for m in range(M)])
        lut = _native.make_lut(alpha)
        b, _ = rand_seqs(N, L, alpha, seed=L)
        b[N - 100:] = b[:100]
        nm, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
        assert np.isfinite(nm).all() and np.array_equal(mean, np.mean(nm, axis=1))
        assert np.array_equal(nm[N - 100:], nm[:100])
        sample = np.random.default_rng(L).choice(N, 160, replace=False)
        codes = lut[b[sample]]
        for m in range(M):
            assert_scores(nm[sample, m], c_oracle.forward("cnn", codes, 20, ws[m]), f"L={L} member {m}")
        part, _ = eng.score(list(natives), b[1_234:5_678], lut)
        assert np.array_equal(part, nm[1_234:5_678])


# ------------------------------------------------------------------ randomised shapes (dispatch boundaries)
def test_random_shapes_against_oracle(eng):
    """60 random (kind, L, alphabet, F, H, K, N) draws straddling the MFMA / shape-agnostic dispatch
    boundaries (alphabets of 2..21 letters, hidden sizes 1..230, odd filter counts, even kernels)."""
    rng = np.random.default_rng(2026)
    letters = "ACDEFGHIKLMNPQRSTVWYX"
    for trial in range(60):
        kind = ("cnn", "mlp", "ge")[trial % 3]
        A = int(rng.choice([2, 3, 4, 4, 5, 20, 20, 21]))
        alpha = letters[:A]
        L = int(rng.integers(1, 41))
        H = int(rng.choice([1, 7, 16, 33, 64, 100, 100, 128, 130, 200, 230]))
        F = int(rng.choice([1, 8, 32, 32, 32])) if kind == "cnn" else 0
        K = int(rng.integers(1, min(L, 6) + 1)) if kind == "cnn" else 0
        if kind == "cnn" and rng.random() < 0.5 and L >= 5:
            K = 5
        n = int(rng.choice([1, 2, 15, 16, 17, 100, 333]))
        nm, w = make_native(eng, kind, L, A, H, F, K, seed=trial)
        b, seqs = rand_seqs(n, L, alpha, seed=trial)
        got, _ = eng.score([nm], b, _native.make_lut(alpha))
        want = ref_np.keras_fitness(seqs, alpha, kind, w, exact=True)
        assert_scores(got[:, 0], want, f"trial {trial}: {kind} L={L} A={A} F={F} H={H} K={K} n={n}")


@pytest.mark.parametrize("kind,L,A,alpha,M,n", [("cnn", 30, 20, s_utils.AAS, 3, 700), ("cnn", 66, 20, s_utils.AAS, 5, 130),
                                                ("mlp", 90, 20, s_utils.AAS, 3, 900), ("mlp", 14, 4, "UGCA", 5, 4000),
                                                ("ge", 30, 20, s_utils.AAS, 11, 1000), ("cnn", 14, 4, "UGCA", 4, 70001)])
def test_multi_member_launches(eng, kind, L, A, alpha, M, n):
    """Several members in one fused launch: workgroup ranges straddle member boundaries (weights are
    reloaded in LDS mid-range) -- pair kernel, L2-gathered MLP, unrolled L=14 kernel included."""
    F, K = (32, 5) if kind == "cnn" else (0, 0)
    natives, ws = zip(*[make_native(eng, kind, L, A, 100, F, K, seed=300 + m) for m in range(M)])
    b, seqs = rand_seqs(n, L, alpha, seed=M * 7 + L)
    nm, mean = eng.score(list(natives), b, _native.make_lut(alpha), want_matrix=True, want_mean=True)
    for m in range(M):
        assert_scores(nm[:, m], ref_np.keras_fitness(seqs, alpha, kind, ws[m], exact=True), f"{kind} member {m}/{M}")
    assert np.array_equal(mean, np.mean(nm, axis=1))


@pytest.mark.parametrize("L,alpha,F,K,n,M", [(8, "TGCA", 32, 3, 3000, 3), (14, "UGCA", 32, 7, 3000, 2), (50, "UGCA", 24, 3, 40, 1),
                                             (50, "UGCA", 17, 7, 40, 1), (8, "TGCA", 20, 5, 70_000, 3), (9, "TGCA", 32, 7, 100, 1),
                                             (60, s_utils.AAS, 32, 3, 300, 2), (60, s_utils.AAS, 28, 7, 33, 1), (237, s_utils.AAS, 32, 3, 20, 1),
                                             (7, "TGCA", 32, 7, 5, 1), (3, "TGCA", 32, 3, 5, 1),
                                             (14, "UGCA", 16, 5, 3000, 3), (8, "TGCA", 8, 5, 70_000, 2), (50, "UGCA", 16, 5, 40, 1),
                                             (60, s_utils.AAS, 16, 5, 100, 2), (8, "TGCA", 1, 5, 64, 1)])
def test_cnn_other_kernel_sizes_and_filter_counts_on_mfma(eng, L, alpha, F, K, n, M):
    """kernel_size 3 / 7 and num_filters 1..32 run on the MFMA kernels too (one or two channel tiles, zero-padded;
    generic window code): scores vs the oracle, and vs the shape-agnostic kernels of the same launch."""
    natives, ws = zip(*[make_native(eng, "cnn", L, len(alpha), 100, F, K, seed=500 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=L * K + F)
    got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    k = min(n, 200)
    codes = lut[b[:k]]
    for m in range(M):
        assert_scores(got[:k, m], c_oracle.forward("cnn", codes, len(alpha), ws[m]), f"L={L} F={F} K={K}")
    assert np.array_equal(mean, np.mean(got, axis=1))
    try:
        eng.set_option("force_generic", 1)
        ref, _ = eng.score(list(natives), b, lut)
    finally:
        eng.set_option("force_generic", 0)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("L,n,M", [(10, 3000, 2), (5, 100, 1), (64, 20, 3), (30, 70000, 1)])
def test_cnn_binary_alphabet_on_mfma(eng, L, n, M):
    """`BA = "01"` (sequence_utils.py:16): the canonical CNN on a 2-letter alphabet (conv3 has a single tap) runs on the
    MFMA kernels -- bulk, small-batch and position-segmented forms -- and agrees with the oracle and the VALU kernels."""
    pairs = [make_native(eng, "cnn", L, 2, 100, 32, 5, seed=60 + m) for m in range(M)]
    b, seqs = rand_seqs(n, L, "01", seed=L)
    lut = _native.make_lut("01")
    got, _ = eng.score([p[0] for p in pairs], b, lut)
    for m in range(M):
        assert_scores(got[:, m], ref_np.keras_fitness(seqs, "01", "cnn", pairs[m][1], exact=True), f"BA cnn L={L} member {m}")
    eng.set_option("force_generic", 1)
    try:
        gen, _ = eng.score([p[0] for p in pairs], b[:500], lut)
    finally:
        eng.set_option("force_generic", 0)
    assert np.abs(gen - got[:500]).max() <= 2e-5 * np.abs(gen).max() + 2e-6
    bb = b.copy(); bb[n // 2, L - 1] = ord("2")
    with pytest.raises(ValueError):
        eng.score([p[0] for p in pairs], bb, lut)
    # other hidden widths (round 6): conv-only kernel + head kernel instead of the shape-agnostic kernels (39 ms -> 0.3 ms for 1e5 x L=40, H=200)
    for H in (200, 64):
        nm, w = make_native(eng, "cnn", L, 2, H, 32, 5, seed=90 + H)
        k = min(n, 2000)
        got_h, _ = eng.score([nm], b[:k], lut)
        assert_scores(got_h[:, 0], ref_np.keras_fitness(seqs[:k], "01", "cnn", w, exact=True), f"BA cnn L={L} H={H}")
        assert eng.counters()["forwards"] >= 0


def _random_case(seed):
    """One seeded draw of (architecture, shape, members, batch size) from everything the constructors of cnn.py:10-21,
    mlp.py:10-19 and global_epistasis_model.py:15-24 accept -- canonical and odd sizes alike, so that every kernel
    family (fused / conv + head / pair / segmented / quad / dense / slab / shape-agnostic) is hit by some draw."""
    rng = np.random.default_rng(9000 + seed)
    kind = ("cnn", "cnn", "mlp", "ge")[int(rng.integers(0, 4))]
    alpha = ("TGCA", "UGCA", s_utils.AAS, "01", "ACGTN")[int(rng.choice(5, p=[.3, .25, .25, .1, .1]))]
    A = len(alpha)
    K = int(rng.integers(2, 8)) if rng.random() < 0.5 else 5
    F = int(rng.choice([8, 16, 24, 32, 32, 32, 48, 64]))
    H = int(rng.choice([7, 16, 50, 64, 100, 100, 100, 112, 128, 130, 200]))
    lmax = 40 if A == 20 else 70
    L = 8 if rng.random() < 0.25 else int(rng.integers(max(K, 3), lmax + 1))
    M = int(rng.integers(1, 5))
    n = int(rng.choice([1, 2, 15, 16, 17, 33, 100, 257, 1000, 3001]))
    if kind != "cnn":
        F = K = 0
    return kind, alpha, A, L, H, F, K, M, n


@pytest.mark.parametrize("seed", range(64))
def test_random_shapes_against_the_oracle_and_batch_invariance(eng, seed):
    """Seeded sweep over architectures, alphabets, shapes, member counts and batch sizes.  (1) every member's scores match
    the float64 oracle within the stated tolerance; (2) the device mean is np.mean of the matrix, bit for bit;
    (3) BATCH INVARIANCE: a sequence's score does not depend on the call it arrives in -- a prefix scored on its own
    (which may select another kernel form: quad, position-segmented, fewer waves) gives the same bits, as the
    reference's explorers assume when they cache and compare model scores across calls of different sizes."""
    kind, alpha, A, L, H, F, K, M, n = _random_case(seed)
    what = f"seed {seed}: {kind} alphabet {alpha!r} L={L} H={H} F={F} K={K} M={M} n={n}"
    natives, ws = zip(*[make_native(eng, kind, L, A, H, F, K, seed=500 + 7 * seed + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=seed)
    got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    for m in range(M):
        assert_scores(got[:, m], ref_np.keras_fitness(seqs, alpha, kind, ws[m], exact=True), what + f" member {m}")
    assert np.array_equal(mean, np.mean(got, axis=1)), what
    rng = np.random.default_rng(seed)
    for k in sorted({1, min(n, 20), int(rng.integers(1, n + 1))}):
        part, _ = eng.score(list(natives), b[:k], lut)
        assert np.array_equal(part, got[:k]), what + f": prefix of {k} scored differently"
    # a character outside the alphabet anywhere in the batch fails the call (sequence_utils.py:46)
    bad = b.copy()
    bad[int(rng.integers(0, n)), int(rng.integers(0, L))] = ord("#")
    with pytest.raises(ValueError):
        eng.score(list(natives), bad, lut)
