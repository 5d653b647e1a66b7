"""CPU checks of the pieces of the HIP path that do not need a GPU:
  * the host weight packing (Keras order -> MFMA fragment order) + the kernels'
    dataflow, via the lane-level model in tests/mfma_sim.py, against the oracle;
  * the bit-parallel Levenshtein (same header the device kernel compiles)."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import mfma_sim
from flexs_amd import _native
from oracle import c_oracle, ref_np


@pytest.mark.parametrize("L,A,alpha,H", [(8, 4, "TGCA", 100), (5, 4, "TGCA", 100), (14, 4, "UGCA", 100),
                                         (27, 20, ref_np.AAS, 100), (8, 4, "TGCA", 1), (8, 4, "TGCA", 50),
                                         (8, 4, "TGCA", 128), (9, 4, "TGCA", 200), (9, 4, "TGCA", 250)])
def test_cnn_dataflow_matches_oracle(L, A, alpha, H):
    K, F = 5, 32
    rng = np.random.default_rng(L)
    w = ref_np.synth_weights(ref_np.cnn_shapes(L, A, F, H, K), 1000)
    packed = _native.debug_pack_weights(_native.FX_CNN, L, A, F, H, K, w)
    lay = _native.debug_pack_layout(_native.FX_CNN, L, A, F, H, K)
    assert lay["FT"] == 2 and lay["HTR"] == -(-H // 16) and lay["HT"] in (1, 2, 4, 7, 8, 13, 16) and lay["HT"] >= lay["HTR"]
    codes = rng.integers(0, A, (16, L)).astype(np.uint8)
    seqs = ["".join(alpha[c] for c in r) for r in codes]
    want = ref_np.keras_fitness(seqs, alpha, "cnn", w, exact=True)
    for gather in (True, False):
        got = mfma_sim.cnn_tile(packed, lay, codes, A, K, F, H, conv1_gather=gather)
        assert np.abs(got - want).max() < 1e-12


@pytest.mark.parametrize("L,A,alpha", [(27, 20, ref_np.AAS), (5, 20, ref_np.AAS), (12, 4, "TGCA"), (40, 20, ref_np.AAS)])
def test_cnn_pair_dataflow_matches_oracle(L, A, alpha):
    """Two-waves-per-tile form (score_cnn_pair.hip): scatter-form conv3 window + LDS swap of the halves."""
    K, F, H = 5, 32, 100
    rng = np.random.default_rng(L + A)
    w = ref_np.synth_weights(ref_np.cnn_shapes(L, A, F, H, K), 77)
    packed = _native.debug_pack_weights(_native.FX_CNN, L, A, F, H, K, w)
    lay = _native.debug_pack_layout(_native.FX_CNN, L, A, F, H, K)
    codes = rng.integers(0, A, (16, L)).astype(np.uint8)
    seqs = ["".join(alpha[c] for c in r) for r in codes]
    want = ref_np.keras_fitness(seqs, alpha, "cnn", w, exact=True)
    assert np.abs(mfma_sim.cnn_pair_tile(packed, lay, codes, A, K, F, H) - want).max() < 1e-12


@pytest.mark.parametrize("L,A,alpha,H", [(14, 4, "UGCA", 100), (8, 4, "TGCA", 100), (9, 20, ref_np.AAS, 100),
                                         (14, 4, "UGCA", 97), (14, 4, "UGCA", 104), (14, 4, "UGCA", 107), (14, 4, "UGCA", 112),
                                         (14, 4, "UGCA", 7), (14, 4, "UGCA", 33), (14, 4, "UGCA", 64), (14, 4, "UGCA", 120),
                                         (14, 4, "UGCA", 200), (14, 4, "UGCA", 208), (14, 4, "UGCA", 256)])
def test_mlp_ge_dataflow_matches_oracle(L, A, alpha, H):
    rng = np.random.default_rng(7)
    codes = rng.integers(0, A, (16, L)).astype(np.uint8)
    seqs = ["".join(alpha[c] for c in r) for r in codes]
    for kind, shapes, fn, k in (("mlp", ref_np.mlp_shapes(L, A, H), mfma_sim.mlp_tile, _native.FX_MLP),
                                ("ge", ref_np.ge_shapes(L, A, H), mfma_sim.ge_tile, _native.FX_GE)):
        w = ref_np.synth_weights(shapes, 5)
        packed = _native.debug_pack_weights(k, L, A, 0, H, 0, w)
        lay = _native.debug_pack_layout(k, L, A, 0, H, 0)
        want = ref_np.keras_fitness(seqs, alpha, kind, w, exact=True)
        assert np.abs(fn(packed, lay, codes, A, H) - want).max() < 1e-12, kind
        if kind == "mlp":
            assert np.abs(fn(packed, lay, codes, A, H, l1_gather=False) - want).max() < 1e-12


def test_packed_sizes():
    # CNN(32,100,K5) on a 4-letter alphabet: 25 780 floats = 100.7 KiB -> fits the 160 KiB LDS
    assert _native.lib().fx_debug_packed_size(_native.FX_CNN, 8, 4, 32, 100, 5) == 26500            # (+20 rows of 36 floats: the plain conv1 rows of the gather form, FX_C1_ROW)
    lay = _native.debug_pack_layout(_native.FX_CNN, 237, 20, 32, 100, 5)
    assert lay["conv_floats"] * 4 < 160 * 1024            # conv part alone fits for the protein alphabet
    with pytest.raises(ValueError):
        _native.debug_pack_weights(_native.FX_CNN, 8, 4, 32, 100, 5, [np.zeros(3, np.float32)])


# ---------------------------------------------------------------- bit-parallel Levenshtein
def test_myers_known_answers(golden_dir):
    known = json.load(open(os.path.join(golden_dir, "edit_distance_known.json")))["known"]
    for k in known:                                        # L = 66, 90, 238: 2- and 4-word cases
        assert _native.debug_myers(k["seq"].encode(), k["wt"].encode()) == k["levenshtein_dp"]


@settings(max_examples=300, deadline=None)
@given(st.text(alphabet="ACGT", min_size=0, max_size=40), st.text(alphabet="ACGT", min_size=0, max_size=40))
def test_myers_equals_dp_short(a, b):
    assert _native.debug_myers(a.encode(), b.encode()) == c_oracle.levenshtein(a, b)


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 256), st.integers(0, 300), st.integers(0, 2**31), st.integers(2, 20))
def test_myers_equals_dp_multiword(la, lb, seed, nsym):
    rng = np.random.default_rng(seed)
    a = bytes(rng.integers(65, 65 + nsym, la).astype(np.uint8))
    # make b a noisy copy of a half of the time so that distances are small and structured
    if seed % 2 and la:
        b = bytearray(a)
        for _ in range(int(rng.integers(0, 8))):
            i = int(rng.integers(0, max(len(b), 1)))
            op = int(rng.integers(0, 3))
            if op == 0 and b:
                b[i % len(b)] = int(rng.integers(65, 65 + nsym))
            elif op == 1 and b:
                del b[i % len(b)]
            else:
                b.insert(i % (len(b) + 1), int(rng.integers(65, 65 + nsym)))
        b = bytes(b)
    else:
        b = bytes(rng.integers(65, 65 + nsym, lb).astype(np.uint8))
    assert _native.debug_myers(a, b) == c_oracle.levenshtein(a, b)


def test_bounded_distance_band_against_the_dp_oracle():
    """csrc/myers.h fx_bounded_distance (the banded kernel behind fx_cache_density: `sequence_density` only asks for distances up to
    its radius), host build: min(levenshtein, K + 1) for K = 1 .. 3 on pairs that are 0 .. 5 edits apart (substitutions, insertions,
    deletions anywhere, so also unequal lengths and NUL-padded rows), on unrelated pairs, on empty strings; Hamming likewise."""
    rng = np.random.default_rng(11)
    pairs = [(b"", b""), (b"A", b""), (b"", b"AC"), (b"ACGT", b"ACGT"), (b"ACGT", b"CGTA"), (b"AAAA", b"AAAAAAA"), (b"ACGTACGT", b"TGCATGCA")]
    for _ in range(1500):
        la = int(rng.integers(0, 60))
        nsym = int(rng.integers(2, 21))
        a = bytes(rng.integers(65, 65 + nsym, la).astype(np.uint8))
        b = bytearray(a)
        for _e in range(int(rng.integers(0, 6))):
            op = int(rng.integers(0, 3))
            if op == 0 and b:
                b[int(rng.integers(0, len(b)))] = int(rng.integers(65, 65 + nsym))
            elif op == 1 and b:
                del b[int(rng.integers(0, len(b)))]
            else:
                b.insert(int(rng.integers(0, len(b) + 1)), int(rng.integers(65, 65 + nsym)))
        pairs.append((a, bytes(b)))
        if rng.random() < 0.2:
            pairs.append((a, bytes(rng.integers(65, 65 + nsym, int(rng.integers(0, 60))).astype(np.uint8))))
    for a, b in pairs:
        d = c_oracle.levenshtein(a, b)
        for K in (1, 2, 3):
            assert _native.debug_bounded_distance(a, b, K) == min(d, K + 1), (a, b, K, d)
            assert _native.debug_bounded_distance(b, a, K) == min(d, K + 1), (b, a, K, d)
        if len(a) == len(b):
            h = sum(x != y for x, y in zip(a, b))
            for K in (1, 2, 3):
                assert _native.debug_bounded_distance(a, b, K, hamming=True) == min(h, K + 1)
    assert _native.debug_bounded_distance(b"AC", b"AC", 4) == -1 and _native.debug_bounded_distance(b"AC", b"AC", 0) == -1


def test_myers_boundaries():
    for L in (1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 384, 385, 512, 513, 735, 768):
        a = bytes([65 + (i * 7) % 4 for i in range(L)])
        b = bytes([65 + (i * 5 + 1) % 4 for i in range(L)])
        assert _native.debug_myers(a, b) == c_oracle.levenshtein(a, b)
        assert _native.debug_myers(a, a) == 0
        assert _native.debug_myers(a, a[1:] + a[:1]) == c_oracle.levenshtein(a, a[1:] + a[:1])
    # beyond 768 symbols (12 words) the device runs the recurrence in strips of 768 pattern rows; so does the hook
    assert _native.debug_myers(b"x" * 769, b"x") == 768


def test_myers_in_strips_for_patterns_of_any_length():
    """`editdistance.eval` (noisy_abstract_model.py:51) has no length limit: beyond 768 symbols the kernels run the
    bit-parallel recurrence in strips whose boundary is one horizontal delta per text column (csrc/myers.h
    fx_myers_strip).  Property test against the DP oracle: 64-row strips (many boundaries on short strings, every
    strip-boundary / word-boundary alignment) and the device's 768-row strips on long ones."""
    rng = np.random.default_rng(5)
    for trial in range(300):
        nsym = int(rng.choice([2, 4, 20]))
        la, lb = int(rng.integers(0, 300)), int(rng.integers(0, 300))
        a = bytes(rng.integers(65, 65 + nsym, la).astype(np.uint8))
        if trial % 2 and la:
            b = bytearray(a)
            for _ in range(int(rng.integers(0, 10))):
                i = int(rng.integers(0, max(len(b), 1)))
                op = int(rng.integers(0, 3))
                if op == 0 and b:
                    b[i % len(b)] = int(rng.integers(65, 65 + nsym))
                elif op == 1 and b:
                    del b[i % len(b)]
                else:
                    b.insert(i % (len(b) + 1), int(rng.integers(65, 65 + nsym)))
            b = bytes(b)
        else:
            b = bytes(rng.integers(65, 65 + nsym, lb).astype(np.uint8))
        want = c_oracle.levenshtein(a, b)
        assert _native.debug_myers_strips(a, b, 1) == want, (trial, la, len(b))
        assert _native.debug_myers_strips(a, b, 12) == want
    for la in (63, 64, 65, 128, 129, 767, 768, 769, 1000, 1536, 1537, 2400):
        a = bytes([65 + (i * 7) % 4 for i in range(la)])
        b = bytes([65 + (i * 5 + 1) % 4 for i in range(la + 3)])
        for x, y in ((a, b), (b, a), (a, a), (a, a[5:] + a[:5]), (a, b""), (b"", a)):
            want = c_oracle.levenshtein(x, y)
            assert _native.debug_myers_strips(x, y, 1) == want and _native.debug_myers_strips(x, y, 12) == want
            assert _native.debug_myers(x, y) == want


def test_mlp_pair_rows_are_sums_of_the_single_position_rows():
    """MLP on a 4-letter alphabet: the packed image carries, after the vectors, one row per (pair of positions, pair of
    letters) = float32 sum of the two single-position rows (pack.cpp); an odd last position keeps its four rows.  Rows
    are 4 zero floats apart (LDS bank spread of the 16-row gather)."""
    from oracle import ref_np

    for L, H in ((14, 100), (7, 33), (2, 16)):
        w = ref_np.synth_weights(ref_np.mlp_shapes(L, 4, H), 5)
        lay = _native.debug_pack_layout(_native.FX_MLP, L, 4, 0, H, 0)
        packed = _native.debug_pack_weights(_native.FX_MLP, L, 4, 0, H, 0, w)
        R = 16 * lay["HT"]
        rows = packed[lay["off_w1p"]: lay["off_w1p"] + L * 4 * R].reshape(L * 4, R)
        n_pair_rows = (L // 2) * 16 + (L % 2) * 4
        assert packed.shape[0] == lay["total_floats"] + n_pair_rows * (R + 4)
        padded = packed[lay["total_floats"]:].reshape(n_pair_rows, R + 4)
        pair = padded[:, :R]
        assert not padded[:, R:].any()
        for pi in range(L // 2):
            for c0 in range(4):
                for c1 in range(4):
                    assert np.array_equal(pair[pi * 16 + 4 * c0 + c1], rows[(2 * pi) * 4 + c0] + rows[(2 * pi + 1) * 4 + c1])
        if L % 2:
            assert np.array_equal(pair[(L // 2) * 16:], rows[(L - 1) * 4:])
    # other alphabets have no pair table
    assert _native.lib().fx_debug_packed_size(_native.FX_MLP, 14, 20, 0, 100, 0) == _native.debug_pack_layout(_native.FX_MLP, 14, 20, 0, 100, 0)["total_floats"]
