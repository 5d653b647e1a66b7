"""GPU parity of the alternative launch forms of the scoring kernels (position-split, pair, layer-parallel, quad, split conv / head,
small dense launches, first-layer forms): each against the oracle and, where it claims so, bit for bit against the main kernel."""
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


@pytest.mark.parametrize("L,n,M,H", [(100, 20, 3, 100), (50, 1, 1, 100), (50, 100, 2, 100), (14, 7, 1, 100), (9, 3, 1, 100),
                                      (28, 33, 2, 64), (100, 16, 1, 200), (61, 40, 1, 256), (8, 5, 3, 100), (5, 2, 1, 100)])
def test_cnn_position_split_small_batches(eng, L, n, M, H):
    """Small batches of the 4-letter CNN kernel: the waves of a workgroup split one tile's positions (cnn_seg).
    Forced on, automatic and off must agree bit for bit, and match the oracle -- including sequences with fewer
    conv positions than waves (L = 5, 8, 9)."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, H, 32, 5, seed=70 + m) for m in range(M)])
    lut = _native.make_lut("UGCA")
    b, seqs = rand_seqs(n, L, "UGCA", seed=L * 3 + n)
    try:
        eng.set_option("cnn_seg", 0)
        whole, _ = eng.score(list(natives), b, lut)
        for m in range(M):
            assert_scores(whole[:, m], ref_np.keras_fitness(seqs, "UGCA", "cnn", ws[m], exact=True), f"L={L} H={H}")
        for mode in (1, -1):
            eng.set_option("cnn_seg", mode)
            got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(got, whole), (L, n, M, H, mode)
            assert np.array_equal(mean, np.mean(whole, axis=1))
        eng.set_option("cnn_seg", 1)
        bad = b.copy()
        bad[n // 2, L - 1] = ord("Z")
        with pytest.raises(ValueError):
            eng.score(list(natives), bad, lut)
    finally:
        eng.set_option("cnn_seg", -1)


@pytest.mark.parametrize("L,n,M", [(237, 40, 3), (238, 1, 1), (90, 16, 2), (90, 33, 1), (31, 5, 1), (60, 100, 2)])
def test_cnn_pair_segmented_small_batches(eng, L, n, M):
    """Position-segmented form of the wide-alphabet CNN kernel (CMA-ES / DyNA-PPO sized calls): every forced
    segmentation (SB workgroups per tile) and the automatic one give the whole-sequence form's scores bit for bit,
    and those match the oracle."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 20, 100, 32, 5, seed=40 + m) for m in range(M)])
    lut = _native.make_lut(s_utils.AAS)
    b, seqs = rand_seqs(n, L, s_utils.AAS, seed=L + n)
    try:
        eng.set_option("cnn_pair_seg", 0)
        whole, _ = eng.score(list(natives), b, lut)
        k = min(n, 64)
        for m in range(M):
            assert_scores(whole[:k, m], ref_np.keras_fitness(seqs[:k], s_utils.AAS, "cnn", ws[m], exact=True), f"pair L={L}")
        for sb in (1, 2, 3, 5, 8, -1):
            eng.set_option("cnn_pair_seg", sb)
            got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(got, whole), (L, n, M, sb)
            assert np.array_equal(mean, np.mean(whole, axis=1))
        eng.set_option("cnn_pair_seg", 7)
        bad = b.copy()
        bad[n // 2, L - 3] = ord("Z")                     # bad character inside some segment only
        with pytest.raises(ValueError):
            eng.score(list(natives), bad, lut)
    finally:
        eng.set_option("cnn_pair_seg", -1)


@pytest.mark.parametrize("L,n,M", [(237, 40, 3), (237, 1, 1), (237, 15, 3), (238, 16, 1), (90, 10, 3), (90, 33, 2), (100, 80, 3), (28, 5, 1),
                                   (237, 17, 8), (150, 48, 5)])
def test_cnn_layer_parallel_small_batches(eng, L, n, M):
    """Round 4: small batches of the canonical protein CNN (a CMA-ES population of 15-40, a DyNA-PPO environment batch, one
    sequence) LAYER-PARALLEL over the chip (`cnn_lp`, default on: conv1 + conv2 per position block, conv2 outputs through
    device memory, one grid barrier, conv3 + pool per position block, head by the last block of a tile) instead of position
    segments that recompute a 22-position halo each.  Per output element the pair kernel's MFMA sequence, so the SAME BITS as
    the whole-sequence walk and as the segmented form; beside the float64 oracle; repeated launches (the barrier counter only
    ever grows), a bad character in some block only, and batches too large for one wave of the grid keep the old forms."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 20, 100, 32, 5, seed=80 + m) for m in range(M)])
    lut = _native.make_lut(s_utils.AAS)
    b, seqs = rand_seqs(n, L, s_utils.AAS, seed=3 * L + n)
    try:
        eng.set_option("cnn_pair_seg", 0)
        whole, _ = eng.score(list(natives), b, lut)
        eng.set_option("cnn_pair_seg", -1)
        eng.set_option("cnn_lp", 0)
        seg, _ = eng.score(list(natives), b, lut)
        eng.set_option("cnn_lp", 1)
        assert np.array_equal(seg, whole)
        for rep in range(4):
            got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(got, whole), (L, n, M, rep)
            assert np.array_equal(mean, np.mean(whole, axis=1))
        for cut in (1, 16, 17, n - 1):                                    # batch invariance across forms
            if 0 < cut < n:
                part, _ = eng.score(list(natives), b[:cut], lut)
                assert np.array_equal(part, whole[:cut]), cut
        k = min(n, 48)
        for m in range(M):
            assert_scores(whole[:k, m], ref_np.keras_fitness(seqs[:k], s_utils.AAS, "cnn", ws[m], exact=True), f"lp L={L} member {m}")
        for where in ((0, 0), (n // 2, L // 2), (n - 1, L - 1)):          # a bad character that only one position block reads
            bad = b.copy()
            bad[where] = ord("Z")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, whole)
    finally:
        eng.set_option("cnn_pair_seg", -1)
        eng.set_option("cnn_lp", 1)


# ------------------------------------------------------------------ ensembles
@pytest.mark.parametrize("L,alpha,H,M,n", [(90, s_utils.AAS, 100, 8, 3001), (90, s_utils.AAS, 100, 1, 17), (8, "TGCA", 100, 3, 1000),
                                           (33, s_utils.AAS, 50, 2, 100), (100, "UGCA", 128, 1, 257), (64, s_utils.AAS, 16, 5, 16)])
def test_ge_byte_table_first_layer(eng, L, alpha, H, M, n):
    """GlobalEpistasis layer 1 gathered from the per-position table indexed by the raw byte (LDS-resident, padding
    rows of zeros, bytes of the following rows read on full trips) gives the bits of the LUT + code-indexed gather --
    same summation order -- and the oracle's values; characters are validated by the first member's units only."""
    A = len(alpha)
    pairs = [make_native(eng, "ge", L, A, H, seed=1000 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=5)
    got, _ = eng.score(nms, b, lut)
    eng.set_option("ge_bytetab", 0)
    try:
        old, _ = eng.score(nms, b, lut)
    finally:
        eng.set_option("ge_bytetab", 1)
    assert np.array_equal(got, old)
    for m in (0, M - 1):
        assert_scores(got[:, m], ref_np.keras_fitness(seqs, alpha, "ge", pairs[m][1], exact=True), f"ge byte table member {m}")
    # a bad character anywhere (last row, last position; first row) is reported whichever member's units meet it
    for r, c in ((n - 1, L - 1), (0, 0), (n // 2, L // 2)):
        bb = b.copy()
        bb[r, c] = ord("z")
        with pytest.raises(ValueError):
            eng.score(nms, bb, lut)
    # ... and a NaN weight is NOT a bad character: np.nan_to_num semantics (keras_model.py:77)
    w = [x.copy() for x in pairs[0][1]]
    w[0][3, 0] = np.nan
    nms[0].set_weights(w)
    out, _ = eng.score(nms, b, lut)                                   # must not raise
    clean = b[:, 0] != ord(alpha[3])                                   # rows that never touch the NaN weight
    assert np.array_equal(out[clean], got[clean]) and not np.isnan(out).any()
    # a different alphabet order for the same model rebuilds the table
    alpha2 = alpha[::-1]
    lut2 = _native.make_lut(alpha2)
    nms[0].set_weights(pairs[0][1])
    got2, _ = eng.score(nms[:1], b, lut2)
    assert_scores(got2[:, 0], ref_np.keras_fitness(seqs, alpha2, "ge", pairs[0][1], exact=True), "reversed alphabet")


@pytest.mark.parametrize("L,H,M,n", [(14, 100, 1, 5000), (9, 100, 3, 333), (8, 64, 2, 100), (2, 100, 1, 40), (50, 100, 1, 1000),
                                     (14, 200, 1, 40_000), (15, 200, 2, 9_000), (8, 256, 1, 33_000), (16, 200, 1, 9_000), (1, 200, 1, 5_000)])
def test_mlp_pair_rows_first_layer(eng, L, H, M, n):
    """MLP layer 1 on a 4-letter alphabet from the pre-summed pair rows: within tolerance of the oracle and of the
    row-per-position gather (one extra float32 rounding per pair), odd lengths, bad characters in either half of a pair.
    H > 128 (round 6): the slab form keeps the pair rows in LDS beside its slabs where they fit (seq_len <= 15 at H = 200),
    the plain rows otherwise (L = 16)."""
    pairs = [make_native(eng, "mlp", L, 4, H, seed=70 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut("UGCA")
    b, seqs = rand_seqs(n, L, "UGCA", seed=L)
    got, _ = eng.score(nms, b, lut)
    eng.set_option("mlp_pair", 0)
    try:
        single, _ = eng.score(nms, b, lut)
    finally:
        eng.set_option("mlp_pair", 1)
    for m in range(M):
        want = ref_np.keras_fitness(seqs, "UGCA", "mlp", pairs[m][1], exact=True)
        assert_scores(got[:, m], want, f"mlp pair rows L={L} member {m}")
        assert_scores(single[:, m], want, f"mlp single rows L={L} member {m}")
    for col in {0, min(1, L - 1), L - 1}:
        bb = b.copy()
        bb[n - 1, col] = ord("T")                     # not in "UGCA"
        with pytest.raises(ValueError):
            eng.score(nms, bb, lut)


@pytest.mark.parametrize("kind,L,alpha", [("mlp", 14, "UGCA"), ("ge", 90, s_utils.AAS), ("mlp", 9, "TGCA"), ("ge", 33, s_utils.AAS), ("cnn", 8, "TGCA")])
def test_tile_bytes_staged_through_lds_equal_byte_loads_at_any_alignment(eng, kind, L, alpha):
    """The MLP / GE kernels copy a tile's 16 x L bytes into LDS with 16-byte loads: same bits as the byte-load form
    (`stage_bytes` = 0), for a device buffer that starts at any byte offset (a row offset into a caller's batch),
    batches that end inside a tile, and one-sequence batches."""
    import torch

    F, K = (32, 5) if kind == "cnn" else (0, 0)
    nm, w = make_native(eng, kind, L, len(alpha), 100, F, K, seed=3)
    lut = _native.make_lut(alpha)
    for n, off in ((1000, 0), (1000, 3), (37, 1), (16, 5), (1, 7), (4097, 13)):
        b, seqs = rand_seqs(n, L, alpha, seed=n + off)
        buf = torch.zeros(n * L + 64, dtype=torch.uint8, device="cuda")
        buf[off:off + n * L] = torch.from_numpy(b.reshape(-1)).cuda()
        outs = []
        for stage in (1, 0):
            eng.set_option("stage_bytes", stage)
            out = torch.full((n, 1), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            eng.score_dev([nm], buf.data_ptr() + off, n, L, lut, out.data_ptr(), None)
            eng.sync()
            outs.append(out.cpu().numpy()[:, 0])
        eng.set_option("stage_bytes", 1)
        assert np.array_equal(outs[0], outs[1]), (kind, n, off)
        assert_scores(outs[0], ref_np.keras_fitness(seqs, alpha, kind, w, exact=True), f"{kind} staged n={n} off={off}")


@pytest.mark.parametrize("L,n,M", [(8, 1, 1), (8, 16, 1), (8, 17, 3), (8, 100, 3), (8, 1000, 1), (8, 4000, 3), (8, 10_000, 1), (8, 12_289, 1), (8, 2001, 8),
                                    (14, 1, 1), (14, 20, 3), (14, 1000, 3), (14, 4100, 1), (14, 8200, 1), (16, 33, 2), (5, 50, 1), (6, 700, 3), (11, 257, 2),
                                    (13, 2001, 8)])
@pytest.mark.parametrize("K", [5])
def test_cnn_quad_form_is_bit_identical_to_the_one_wave_kernel(eng, L, n, M, K):
    """Small launches of the canonical 4-letter CNN at seq_len <= 16 (TF-binding 8, RNA 14): a tile shared by four waves,
    each taking every fourth conv position, activations exchanged through LDS layer by layer (score_cnn_quad.hip).  Every output element sees the one-wave kernel's MFMA sequence, so
    the scores are the SAME BITS (a sequence must score alike in a call of 20 and in a batch of 1e5), at any size when
    forced, and a character outside the alphabet is reported from whichever wave reads it."""
    pairs = [make_native(eng, "cnn", L, 4, 100, 32, K, seed=80 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut("TGCA")
    b, seqs = rand_seqs(n, L, "TGCA", seed=n)
    outs = {}
    for mode in (0, 1, 2):
        eng.set_option("cnn_quad", mode)
        try:
            outs[mode], mean = eng.score(nms, b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[mode], axis=1))
        finally:
            eng.set_option("cnn_quad", 1)
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
    # the weights through registers instead of the direct global -> LDS copies (the head then lands before the first
    # tile instead of during its convolutions): same bits; and a byte buffer that is not 4-byte aligned (the first
    # round then reads its bytes from global memory like the later ones)
    eng.set_option("dma_fill", 0)
    try:
        for mode in (1, 2):
            eng.set_option("cnn_quad", mode)
            assert np.array_equal(eng.score(nms, b, lut, want_matrix=True)[0], outs[0])
    finally:
        eng.set_option("dma_fill", 1)
        eng.set_option("cnn_quad", 1)
    import torch
    dev = torch.zeros(n * L + 8, dtype=torch.uint8, device="cuda")
    for shift in (1, 4):
        dev[shift:shift + n * L] = torch.from_numpy(b.reshape(-1)).cuda()
        stride = (n + 3) // 4 * 4
        planes = torch.full((M, stride), float("nan"), device="cuda")
        torch.cuda.synchronize()
        eng.score_planes_dev(nms, dev.data_ptr() + shift, n, L, lut, planes.data_ptr(), stride)
        eng.sync()
        assert np.array_equal(planes[:, :n].cpu().numpy().T, outs[0]), shift
    assert_scores(outs[2][:, M - 1], ref_np.keras_fitness(seqs, "TGCA", "cnn", pairs[M - 1][1], exact=True), f"quad L={L} n={n} M={M}")
    eng.set_option("cnn_quad", 2)
    try:
        for col in (0, L // 2, L - 1):
            bb = b.copy()
            bb[n - 1, col] = ord("U")
            with pytest.raises(ValueError):
                eng.score(nms, bb, lut)
    finally:
        eng.set_option("cnn_quad", 1)
    # hidden sizes whose last tile holds 1 .. 16 units (k-step tail), through the Python API
    for H in (97, 100, 104, 112):
        model = bm.CNN(L, 32, H, "TGCA", kernel_size=K, seed=H)
        got = model.get_fitness(seqs[:50])
        eng.set_option("cnn_quad", 0)
        try:
            assert np.array_equal(model.get_fitness(seqs[:50]), got)
        finally:
            eng.set_option("cnn_quad", 1)


@pytest.mark.parametrize("L,n,M,K", [(8, 20, 3, 3), (14, 100, 3, 3), (14, 20, 1, 7), (16, 1000, 2, 7), (7, 17, 1, 7), (9, 4000, 1, 3), (14, 8000, 1, 3)])
def test_cnn_quad_form_other_kernel_sizes(eng, L, n, M, K):
    """The quad form for kernel sizes 3 and 7 (the other fused instantiations of the one-wave kernel): same bits, oracle."""
    pairs = [make_native(eng, "cnn", L, 4, 100, 32, K, seed=90 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut("UGCA")
    b, seqs = rand_seqs(n, L, "UGCA", seed=n + K)
    outs = {}
    for mode in (0, 1, 2):
        eng.set_option("cnn_quad", mode)
        try:
            outs[mode], _ = eng.score(nms, b, lut, want_matrix=True)
        finally:
            eng.set_option("cnn_quad", 1)
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
    for m in range(M):
        assert_scores(outs[2][:, m], ref_np.keras_fitness(seqs, "UGCA", "cnn", pairs[m][1], exact=True), f"quad K={K} L={L} n={n}")
    bb = b.copy()
    bb[n // 2, L - 1] = ord("T")
    with pytest.raises(ValueError):
        eng.score(nms, bb, lut)


@pytest.mark.parametrize("L,n,M,H", [(8, 20, 3, 10), (8, 700, 1, 16), (14, 100, 3, 30), (14, 20, 1, 50), (16, 1000, 2, 64), (8, 4000, 1, 64), (14, 33, 2, 70),
                                     (8, 100, 3, 90), (14, 5000, 1, 32), (7, 17, 1, 96)])
def test_cnn_quad_form_other_hidden_sizes(eng, L, n, M, H):
    """The quad form for hidden layers of 1 / 2 / 4 tiles (<= 64 units) and for 65-96 units padded to 7 tiles: same bits as
    the one-wave kernel, oracle."""
    pairs = [make_native(eng, "cnn", L, 4, H, 32, 5, seed=95 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut("UGCA")
    b, seqs = rand_seqs(n, L, "UGCA", seed=n + H)
    outs = {}
    for mode in (0, 1, 2):
        eng.set_option("cnn_quad", mode)
        try:
            outs[mode], _ = eng.score(nms, b, lut, want_matrix=True)
        finally:
            eng.set_option("cnn_quad", 1)
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
    for m in range(M):
        assert_scores(outs[2][:, m], ref_np.keras_fitness(seqs, "UGCA", "cnn", pairs[m][1], exact=True), f"quad H={H} L={L} n={n}")


@pytest.mark.parametrize("L,F,H,K,n,M", [(8, 32, 50, 3, 5000, 3), (14, 32, 128, 7, 3000, 2), (14, 32, 200, 3, 2000, 1), (20, 32, 256, 6, 500, 1),
                                         (9, 8, 20, 4, 300, 2), (8, 16, 64, 5, 70_000, 2), (30, 24, 100, 2, 100, 1), (12, 32, 100, 6, 33, 1),
                                         (50, 32, 30, 3, 17, 1), (6, 1, 1, 2, 5, 1), (3, 32, 100, 3, 4, 1), (100, 12, 257 - 1, 4, 64, 1),
                                         (8, 64, 100, 5, 20_000, 3), (14, 48, 100, 3, 1000, 2), (14, 64, 200, 4, 300, 1), (30, 40, 64, 2, 65, 1),
                                         (8, 64, 100, 7, 50, 1)])
def test_cnn_split_conv_and_head_path(eng, L, F, H, K, n, M):
    """CNN shapes without a fused instantiation (kernel_size 2..7 x any hidden width <= 256 x num_filters <= 32, 4-letter
    alphabets) run as conv kernel + head kernel on MFMA: scores vs the oracle and vs the shape-agnostic kernels, the
    mean-only (planes) form, and a bad character."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, H, F, K, seed=900 + m) for m in range(M)])
    lut = _native.make_lut("TGCA")
    b, _ = rand_seqs(n, L, "TGCA", seed=L * 7 + K + H)
    got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    k = min(n, 300)
    for m in range(M):
        assert_scores(got[:k, m], c_oracle.forward("cnn", lut[b[:k]], 4, ws[m]), f"L={L} F={F} H={H} K={K}")
    assert np.array_equal(mean, np.mean(got, axis=1))
    _, mean_only = eng.score(list(natives), b, lut, want_matrix=False, want_mean=True)
    assert np.array_equal(mean_only, mean)
    try:
        eng.set_option("force_generic", 1)
        ref, _ = eng.score(list(natives), b, lut)
    finally:
        eng.set_option("force_generic", 0)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-6)
    bad = b.copy(); bad[n // 2, L // 2] = ord("N")
    with pytest.raises(ValueError):
        eng.score(list(natives), bad, lut)


@pytest.mark.parametrize("A,alpha,L,F,H,K,n,M", [(4, "UGCA", 100, 32, 50, 3, 20, 3), (4, "UGCA", 50, 48, 200, 4, 1, 1), (4, "TGCA", 64, 16, 128, 2, 100, 2),
                                                 (4, "UGCA", 40, 32, 64, 5, 33, 1), (20, s_utils.AAS, 237, 32, 50, 3, 40, 3),
                                                 (20, s_utils.AAS, 90, 24, 200, 4, 1, 1), (20, s_utils.AAS, 120, 32, 128, 6, 16, 2)])
def test_cnn_split_path_position_segmented_small_batches(eng, A, alpha, L, F, H, K, n, M):
    """Small batches of long sequences on the conv + head path (non-canonical CNN shapes): the conv kernel cuts a
    tile's positions over the waves of a workgroup (4-letter alphabets) or over several workgroups (protein alphabet,
    segment maxima meeting in a zeroed pool through atomicMax on the float bits) -- same bits as the whole-sequence
    walk, and the oracle's values."""
    natives, ws = zip(*[make_native(eng, "cnn", L, A, H, F, K, seed=40 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=L + K)
    got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    eng.set_option("cnn_seg", 0)
    eng.set_option("cnn_pair_seg", 0)
    try:
        whole, _ = eng.score(list(natives), b, lut)
    finally:
        eng.set_option("cnn_seg", -1)
        eng.set_option("cnn_pair_seg", -1)
    assert np.array_equal(got, whole)
    for m in range(M):
        assert_scores(got[:, m], ref_np.keras_fitness(seqs, alpha, "cnn", ws[m], exact=True), f"segmented split A={A} L={L} K={K} H={H}")
    assert np.array_equal(mean, np.mean(got, axis=1))
    if A == 20:
        for sb in (1, 2, 3):                        # forced workgroups per tile
            eng.set_option("cnn_pair_seg", sb)
            try:
                forced, _ = eng.score(list(natives), b, lut)
            finally:
                eng.set_option("cnn_pair_seg", -1)
            assert np.array_equal(forced, whole), sb
    bad = b.copy(); bad[n - 1, L - 1] = ord("!")
    with pytest.raises(ValueError):
        eng.score(list(natives), bad, lut)


@pytest.mark.parametrize("L,F,H,K,n,M", [(30, 32, 100, 3, 200, 2), (60, 32, 50, 7, 64, 1), (25, 24, 200, 4, 100, 1), (90, 32, 100, 6, 40, 3),
                                         (237, 32, 64, 3, 17, 1), (8, 32, 256, 2, 33, 1)])
def test_cnn_split_path_protein_alphabet(eng, L, F, H, K, n, M):
    """The conv + head split with the two-waves-per-tile conv kernel (20-letter alphabet, kernel_size 2..7, any hidden
    width): scores vs the oracle and vs the shape-agnostic kernels."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 20, H, F, K, seed=950 + m) for m in range(M)])
    lut = _native.make_lut(s_utils.AAS)
    b, _ = rand_seqs(n, L, s_utils.AAS, seed=L + K + H)
    got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
    for m in range(M):
        assert_scores(got[:, m], c_oracle.forward("cnn", lut[b], 20, ws[m]), f"protein L={L} F={F} H={H} K={K}")
    assert np.array_equal(mean, np.mean(got, axis=1))
    try:
        eng.set_option("force_generic", 1)
        ref, _ = eng.score(list(natives), b, lut)
    finally:
        eng.set_option("force_generic", 0)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-6)
    bad = b.copy(); bad[n // 2, L - 1] = ord("B")
    with pytest.raises(ValueError):
        eng.score(list(natives), bad, lut)


@pytest.mark.parametrize("L,alpha,H,n,M", [(14, "UGCA", 100, 20, 3), (14, "UGCA", 100, 1, 1), (8, "TGCA", 100, 100, 3), (15, "UGCA", 100, 33, 2), (50, "UGCA", 100, 17, 1),
                                           (100, "UGCA", 100, 400, 3), (14, "UGCA", 200, 20, 3), (50, "UGCA", 200, 100, 1), (30, s_utils.AAS, 100, 40, 2),
                                           (90, s_utils.AAS, 100, 20, 3), (237, s_utils.AAS, 100, 16, 1), (90, s_utils.AAS, 200, 7, 2), (14, "UGCA", 97, 50, 1),
                                           (14, "UGCA", 112, 1000, 2), (9, "ACGTN", 100, 64, 1), (14, "UGCA", 10, 20, 2), (14, "UGCA", 16, 300, 1),
                                           (20, "UGCA", 30, 17, 3), (14, "UGCA", 50, 100, 1), (33, s_utils.AAS, 64, 20, 2), (14, "UGCA", 128, 40, 1),
                                           (14, "UGCA", 130, 20, 2), (40, "UGCA", 256, 33, 1)])
@pytest.mark.parametrize("kind", ["mlp", "ge"])
def test_mlp_small_launch_form_is_bit_identical_to_the_persistent_kernel(eng, kind, L, alpha, H, n, M):
    """Explorer-size MLP launches: one tile per workgroup, its output tiles dealt to 8 waves, weights read straight from L2
    (score_dense_small.hip).  Same terms in the same order as the persistent kernel -- the pre-summed pair rows where that
    kernel uses them (4-letter alphabets whose table fits LDS), plain rows otherwise, the slab-streamed wide hidden
    layers -- so the SAME BITS, at any size when forced; oracle; a bad character anywhere fails the call."""
    A = len(alpha)
    pairs = [make_native(eng, kind, L, A, H, seed=60 + m) for m in range(M)]
    nms = [p[0] for p in pairs]
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=n + L)
    outs = {}
    for mode in (0, 1, 2):
        eng.set_option("dense_small", mode)
        try:
            outs[mode], mean = eng.score(nms, b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[mode], axis=1))
        finally:
            eng.set_option("dense_small", 1)
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
    for m in range(M):
        assert_scores(outs[2][:, m], ref_np.keras_fitness(seqs, alpha, kind, pairs[m][1], exact=True), f"{kind} small L={L} H={H} n={n}")
    for form in (("mlp_pair", 0), ("ge_bytetab", 0)):         # the plain-row / LUT-indexed first layers on both sides
        eng.set_option(*form)
        try:
            eng.set_option("dense_small", 0)
            ref0, _ = eng.score(nms, b, lut, want_matrix=True)
            eng.set_option("dense_small", 2)
            got0, _ = eng.score(nms, b, lut, want_matrix=True)
            assert np.array_equal(ref0, got0) and np.array_equal(ref0, outs[0]) or form[0] == "mlp_pair"
        finally:
            eng.set_option(form[0], 1)
            eng.set_option("dense_small", 1)
    eng.set_option("dense_small", 2)
    try:
        bb = b.copy()
        bb[n - 1, L - 1] = ord("#")
        with pytest.raises(ValueError):
            eng.score(nms, bb, lut)
    finally:
        eng.set_option("dense_small", 1)


@pytest.mark.parametrize("kind,L,alpha,H,M,n", [
    ("mlp", 14, "UGCA", 100, 1, 100_000), ("mlp", 14, "UGCA", 100, 3, 20_000), ("mlp", 8, "TGCA", 100, 1, 5_000),
    ("mlp", 9, "UGCA", 100, 2, 4_099), ("mlp", 16, "UGCA", 100, 1, 70_001), ("mlp", 17, "UGCA", 100, 1, 9_000),
    ("mlp", 14, "UGCA", 128, 2, 9_001), ("mlp", 14, "UGCA", 112, 1, 6_000), ("mlp", 4, "TGCA", 100, 1, 4_500),
    ("ge", 90, s_utils.AAS, 100, 8, 100_000), ("ge", 90, s_utils.AAS, 100, 1, 100_003), ("ge", 14, "UGCA", 100, 1, 20_000),
    ("ge", 8, "TGCA", 100, 3, 10_000), ("ge", 64, s_utils.AAS, 100, 2, 8_191), ("ge", 100, "UGCA", 100, 1, 6_007),
    ("ge", 128, s_utils.AAS, 128, 1, 5_000), ("ge", 33, s_utils.AAS, 128, 2, 7_000), ("ge", 96, "UGCA", 112, 1, 4_200),
])
def test_software_pipelined_dense_form_gives_the_same_bits(eng, kind, L, alpha, H, M, n):
    """Round 3: the MLP (pair rows) / GlobalEpistasis (byte table) launches run tile t + 1's first layer inside tile t's
    MFMA layers (`dense_pipe` = 1: 8 waves, two-part direct LDS fill).  Every output element sees the arithmetic of the
    round-2 form (`dense_pipe` = 0), so the scores are the SAME BITS -- ragged last tiles, members, any alignment -- and
    both agree with the oracle; a character outside the alphabet is reported from the pipelined first layer too."""
    if not ab_option(eng, "dense_pipe", 1):
        pytest.skip("the software-pipelined dense form (measured 11-13 % slower) lives in the A/B build: make ab")
    eng.set_option("dense_pipe", 0)
    A = len(alpha)
    natives, ws = zip(*[make_native(eng, kind, L, A, H, seed=700 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=L + H + M)
    eng.set_option("dense_small", 0)                      # the persistent kernels at every size
    try:
        outs = {}
        for pipe in (2, 1, 0):                            # 2 / 1: the pipelined form with / without hand-placed operand prefetch
            eng.set_option("dense_pipe", pipe)
            outs[pipe], _ = eng.score(list(natives), b, lut)
            for cut in (1, 16, 17, 4097):                 # batch invariance: prefixes, ragged or not
                if cut < n:
                    part, _ = eng.score(list(natives), b[:cut], lut)
                    assert np.array_equal(part, outs[pipe][:cut]), (pipe, cut)
        assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
        k = min(n, 400)
        for m in range(M):
            assert_scores(outs[1][:k, m], c_oracle.forward(kind, lut[b[:k]], A, ws[m]), f"{kind} L={L} H={H} member {m}")
        eng.set_option("dense_pipe", 1)                   # (the form is optional: measured slower)
        for where in (0, n // 2 + 5, n - 1):              # first tile of a wave, a pipelined tile, the ragged tail
            bad = b.copy(); bad[where, L - 1] = ord("!")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[1])
    finally:
        eng.set_option("dense_pipe", 0)
        eng.set_option("dense_small", 1)


@pytest.mark.parametrize("L,alpha,M", [(8, "TGCA", 3), (8, "TGCA", 2), (14, "UGCA", 3), (8, "TGCA", 8), (14, "UGCA", 16), (8, "TGCA", 7)])
def test_small_launch_fused_ensemble_mean(eng, L, alpha, M):
    """Optional form (`fuse_mean` = 1; measured no faster than the separate 3 us launch, so off by default): explorer-size
    calls of a CNN ensemble average in the scoring kernel itself (the member whose workgroup
    finishes a tile last reads all members' scores back and averages in NumPy's order) instead of launching the mean kernel:
    the same bits as the separate launch and as np.mean of the stacked matrix, for every batch size the small-launch form
    serves, repeated calls (the tickets clean up after themselves), and a bad character still raises."""
    if not ab_option(eng, "fuse_mean", 1):
        pytest.skip("the in-kernel ensemble mean of explorer-size launches (no faster than the mean launch) lives in the A/B build: make ab")
    eng.set_option("fuse_mean", 0)
    members = [bm.CNN(L, 32, 100, alpha, seed=s) for s in range(M)]
    ens = flexs_amd.Ensemble(members)
    stack = flexs_amd.Ensemble(members, combine_with=lambda x: x)
    for n in (1, 5, 16, 17, 20, 33, 48, 100, 400, 2001):
        b, seqs = rand_seqs(n, L, alpha, seed=n)
        want = np.mean(stack.get_fitness(seqs), axis=1)
        for fuse in (1, 0, 1):
            eng.set_option("fuse_mean", fuse)
            try:
                got = ens.get_fitness(seqs)
                natives = [m.native() for m in members]
                _, dev_mean = eng.score(natives, b, members[0]._lut, want_matrix=False, want_mean=True)
            finally:
                eng.set_option("fuse_mean", 0)
            assert np.array_equal(got, want) and np.array_equal(dev_mean, want), (n, fuse)
    eng.set_option("fuse_mean", 1)
    try:
        with pytest.raises(ValueError):
            ens.get_fitness(seqs[:7] + ["Z" * L])
        assert np.array_equal(ens.get_fitness(seqs[:20]), want[:20])
    finally:
        eng.set_option("fuse_mean", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("L,M", [(90, 3), (237, 2), (33, 8), (40, 16)])
def test_host_side_mean_of_small_launched_calls(eng, L, M):
    """Launched mean-only host calls of at most `host_mean_below` sequences (the protein CNN's explorer-size calls): the member
    planes are written straight to pinned host memory and np.mean over the members is taken on the host in NumPy's order -- the
    SAME BITS as the mean kernel (host_mean_below = 0) and as np.mean over the stacked member scores; beside the oracle."""
    members = [bm.CNN(L, 32, 100, s_utils.AAS, seed=200 + s) for s in range(M)]
    ens = flexs_amd.Ensemble(members)
    stack = flexs_amd.Ensemble(members, combine_with=lambda x: x)
    try:
        for n in (1, 5, 16, 40, 256, 257):
            seqs = rand_seqs(n, L, s_utils.AAS, seed=77 + n)[1]
            eng.set_option("host_mean_below", 0)
            want = ens.get_fitness(seqs)
            eng.set_option("host_mean_below", 256)
            got = ens.get_fitness(seqs)
            nm = stack.get_fitness(seqs)
            # ... and whether the host polls the kernel's completion flag (default) or waits for the stream
            eng.set_option("done_flag", 0)
            assert np.array_equal(ens.get_fitness(seqs), got) and np.array_equal(stack.get_fitness(seqs), nm), (L, M, n)
            eng.set_option("done_flag", 1)
            for _ in range(3):                               # (back-to-back flagged calls: every one waits for ITS launch)
                assert np.array_equal(ens.get_fitness(seqs), got), (L, M, n)
            assert np.array_equal(got, want), (L, M, n)
            assert np.array_equal(got, np.mean(nm, axis=1)), (L, M, n)
        seqs = rand_seqs(16, L, s_utils.AAS, seed=5)[1]
        got_nm = stack.get_fitness(seqs)
        for m in (0, M - 1):
            ref = ref_np.keras_fitness(seqs, s_utils.AAS, "cnn", [np.asarray(w, np.float64) for w in members[m].model.get_weights()], exact=True)
            assert_scores(got_nm[:, m], ref, f"host-mean call, member {m}, L={L}")
        bad = list(seqs)
        bad[3] = bad[3][:-1] + "!"
        with pytest.raises(ValueError):
            ens.get_fitness(bad)
        assert np.array_equal(ens.get_fitness(seqs), np.mean(got_nm, axis=1))
    finally:
        eng.set_option("host_mean_below", 256)
        eng.set_option("done_flag", 1)


@pytest.mark.parametrize("kind,L,alpha,H,M,n", [
    ("mlp", 14, "UGCA", 100, 1, 100_000), ("mlp", 14, "UGCA", 100, 1, 100_016), ("mlp", 14, "UGCA", 100, 1, 104_096),
    ("mlp", 14, "UGCA", 100, 1, 108_192), ("mlp", 14, "UGCA", 100, 3, 33_333), ("mlp", 8, "TGCA", 100, 2, 50_001),
    ("mlp", 4, "TGCA", 100, 1, 30_000), ("mlp", 16, "UGCA", 64, 1, 70_001), ("mlp", 14, "UGCA", 112, 1, 41_000),
    ("ge", 90, s_utils.AAS, 100, 1, 100_000), ("ge", 90, s_utils.AAS, 100, 1, 104_096), ("ge", 90, s_utils.AAS, 100, 1, 108_200),
    ("ge", 90, s_utils.AAS, 100, 1, 112_300), ("ge", 90, s_utils.AAS, 100, 8, 100_000), ("ge", 14, "UGCA", 100, 3, 33_333),
    ("ge", 8, "TGCA", 50, 1, 21_000), ("ge", 237, s_utils.AAS, 100, 2, 20_480),
])
def test_shared_last_tiles_of_the_dense_kernel_give_the_same_bits(eng, kind, L, alpha, H, M, n):
    """Round 3: the persistent MLP / GlobalEpistasis kernel leaves the (tiles mod 4) last tiles of a workgroup out of its
    per-SIMD shares and walks them with groups of 8 waves (`dense_coop`, score_dense_tile.h) instead of letting one SIMD
    run an extra tile.  Same arithmetic per output element, so the SAME BITS as one wave per tile (`dense_coop` = 0) --
    every remainder (the sizes put 1, 2 and 3 odd tiles into the workgroups), members, ragged batches -- both agree with
    the oracle, and a bad character in a shared tile is still reported."""
    A = len(alpha)
    natives, ws = zip(*[make_native(eng, kind, L, A, H, seed=900 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=L + H + M + 1)
    eng.set_option("dense_small", 0)
    try:
        outs = {}
        for coop in (2, 1, 0):                             # 2: GlobalEpistasis too (measured slower there, so 1 = MLP only)
            eng.set_option("dense_coop", coop)
            outs[coop], _ = eng.score(list(natives), b, lut)
        assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
        eng.set_option("dense_coop", 2)
        k = min(n, 300)
        for m in range(M):
            assert_scores(outs[1][:k, m], c_oracle.forward(kind, lut[b[:k]], A, ws[m]), f"{kind} L={L} H={H} member {m}")
            assert_scores(outs[1][n - k:, m], c_oracle.forward(kind, lut[b[n - k:]], A, ws[m]), f"{kind} L={L} H={H} member {m} tail")
        # a character outside the alphabet anywhere -- the shared tiles are the last ones of each workgroup's range
        ncu = eng.get_option("num_cus")
        tiles = (n + 15) // 16
        for where in (n - 1, 16 * (tiles // ncu) - 1, n // 2):
            bad = b.copy(); bad[min(max(where, 0), n - 1), L - 1] = ord("!")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[1])
    finally:
        eng.set_option("dense_coop", 1)
        eng.set_option("dense_small", 1)


@pytest.mark.parametrize("kind,L,alpha,H,M,n", [
    ("mlp", 14, "UGCA", 200, 1, 100_000), ("mlp", 14, "UGCA", 200, 1, 100_016), ("mlp", 14, "UGCA", 200, 1, 104_096),
    ("mlp", 14, "UGCA", 200, 1, 108_192), ("mlp", 14, "UGCA", 200, 3, 33_333), ("mlp", 8, "TGCA", 256, 2, 50_001),
    ("mlp", 14, "UGCA", 130, 1, 41_000), ("mlp", 14, "UGCA", 200, 1, 37), ("mlp", 14, "UGCA", 200, 1, 4_113),
    ("ge", 90, s_utils.AAS, 200, 1, 100_000), ("ge", 90, s_utils.AAS, 200, 1, 104_096), ("ge", 90, s_utils.AAS, 200, 8, 50_000),
    ("ge", 14, "UGCA", 256, 3, 33_333), ("ge", 237, s_utils.AAS, 200, 2, 20_481),
])
def test_leftover_tiles_of_the_slab_form_give_the_same_bits(eng, kind, L, alpha, H, M, n):
    """Round 6: hidden sizes above 128 (dyna_ppo.py:54's MLP(seq_len, 200, alphabet)) run in lockstep rounds of 8 tiles per workgroup,
    the H x H blocks streamed through LDS slabs once per round; a last round with one live tile cost as much as a full one (1e5
    sequences: 24.4 tiles per workgroup -> a fourth round for one tile in 106 of 256 workgroups).  Up to `dense_slab_coop` (3)
    leftover tiles are now walked by the workgroup's 8 waves together (score_dense_tile.h, blocks straight from L2).  Same arithmetic
    per output element, so the SAME BITS as the lockstep rounds (`dense_slab_coop` = 0) and as any other limit; both agree with the
    oracle; a bad character in a leftover tile is still reported."""
    A = len(alpha)
    natives, ws = zip(*[make_native(eng, kind, L, A, H, seed=600 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=L + H + M + 2)
    eng.set_option("dense_small", 0)
    try:
        outs = {}
        for coop in (0, 1, 3, 7):
            eng.set_option("dense_slab_coop", coop)
            outs[coop], _ = eng.score(list(natives), b, lut)
        for coop in (1, 3, 7):
            assert np.array_equal(outs[coop].view(np.uint32), outs[0].view(np.uint32)), coop
        eng.set_option("dense_slab_coop", 3)
        k = min(n, 300)
        for m in range(M):
            assert_scores(outs[3][:k, m], c_oracle.forward(kind, lut[b[:k]], A, ws[m]), f"{kind} L={L} H={H} member {m} slab")
            assert_scores(outs[3][n - k:, m], c_oracle.forward(kind, lut[b[n - k:]], A, ws[m]), f"{kind} L={L} H={H} member {m} slab tail")
        ncu = eng.get_option("num_cus")
        tiles = (n + 15) // 16
        for where in (n - 1, 16 * (tiles // ncu) - 1, n // 2):      # the leftover tiles are the last ones of each workgroup's range
            bad = b.copy(); bad[min(max(where, 0), n - 1), L - 1] = ord("!")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[3])
    finally:
        eng.set_option("dense_slab_coop", 3)
        eng.set_option("dense_small", 1)


@pytest.mark.parametrize("M,n", [(3, 100_000), (1, 100_000), (3, 99_985), (1, 300_001), (3, 50_000), (8, 100_000), (2, 70_016), (1, 65_552), (3, 1_000_003)])
def test_quad_tail_of_the_persistent_cnn_kernel_gives_the_same_bits(eng, M, n):
    """Round 6: the unrolled seq_len = 8 kernel (BASELINE configs[0] / [1]) leaves the (tiles mod 4) last tiles of a workgroup out of its
    per-SIMD shares and walks them with wave quads in one round behind the main loop (`cnn_quad_tail`, score_cnn_quad_round.h) -- the
    3 x 1e5 headline launch has 18.31 tiles per SIMD and the SIMDs with a 19th tile set its duration.  Same MFMA / add sequence per
    output element: the SAME BITS as one wave per tile throughout (`cnn_quad_tail` = 0); both agree with the oracle at the head and the
    tail of the batch; a bad character in a tail tile is still reported."""
    L, alpha = 8, "TGCA"
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, 100, 32, 5, seed=300 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=n % 89 + M)
    try:
        outs = {}
        for qt in (0, 1):
            eng.set_option("cnn_quad_tail", qt)
            outs[qt], mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[qt], axis=1))
        assert np.array_equal(outs[1].view(np.uint32), outs[0].view(np.uint32))
        k = min(n, 400)
        for m in range(M):
            assert_scores(outs[1][:k, m], c_oracle.forward("cnn", lut[b[:k]], 4, ws[m], F=32, K=5) if False else
                          ref_np.keras_fitness(seqs[:k], alpha, "cnn", ws[m], exact=True), f"cnn L=8 quad tail member {m}")
            assert_scores(outs[1][n - k:, m], ref_np.keras_fitness(seqs[n - k:], alpha, "cnn", ws[m], exact=True), f"cnn L=8 quad tail member {m} tail")
        # a character outside the alphabet in the tiles that end a workgroup's range (the quad round's) and elsewhere
        ncu = eng.get_option("num_cus")
        tiles = (n + 15) // 16
        per_member_wg = max(ncu // M, 1)
        for where in (n - 1, 16 * (tiles // per_member_wg) - 1, 16 * (tiles // per_member_wg) - 17, n // 2):
            bad = b.copy(); bad[min(max(where, 0), n - 1), L - 1] = ord("!")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[1])
    finally:
        eng.set_option("cnn_quad_tail", 1)


@pytest.mark.parametrize("L,M,n", [(8, 3, 100_000), (8, 3, 99_985), (8, 2, 70_016), (8, 7, 40_000), (8, 8, 30_000), (14, 3, 60_001), (10, 3, 50_000),
                                   (8, 3, 5_000), (8, 5, 131_072)])
def test_batch_launch_takes_the_ensemble_mean_itself(eng, L, M, n):
    """Round 6: fx_score_mean_planes_dev (what DistributedEnsemble and bench.py issue for a mean-only step on rows resident in device
    memory) = scores as member-major planes + np.mean(axis=1) of them bit for bit.  A/B build: the wave that finishes the LAST member
    of a tile averages it in the scoring kernel (`fuse_mean_batch`, fx_fused_mean_tile: written-through scores, one relaxed device-scope
    ticket per tile, no fences; fewer than eight members) -- same bits as the mean kernel, launch after launch (the tickets go back to
    zero); measured slower and not in the production library (profiles/r6_fused_mean_ab.log)."""
    import torch

    alpha = "TGCA"
    natives, _ = zip(*[make_native(eng, "cnn", L, 4, 100, 32, 5, seed=500 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, _ = rand_seqs(n, L, alpha, seed=n % 97 + M)
    nm, _ = eng.score(list(natives), b, lut, want_matrix=True)
    want = np.mean(nm, axis=1)
    d_in = torch.from_numpy(b).cuda()
    stride = (n + 63) // 64 * 64
    try:
        for fuse in (1, 0, 1):
            if not ab_option(eng, "fuse_mean_batch", fuse):
                continue                                             # (the production library has the mean kernel only)
            for rep in range(3 if fuse else 1):
                planes = torch.full((M, stride), float("nan"), device="cuda")
                d_mean = torch.full((n,), float("nan"), device="cuda")
                torch.cuda.synchronize()
                eng.score_mean_planes_dev(list(natives), d_in.data_ptr(), n, L, lut, planes.data_ptr(), stride, d_mean.data_ptr())
                eng.sync()
                assert np.array_equal(planes[:, :n].t().cpu().numpy().view(np.uint32), nm.view(np.uint32)), (fuse, rep)
                assert np.array_equal(d_mean.cpu().numpy().view(np.uint32), want.view(np.uint32)), (fuse, rep)
            d_mean = torch.full((n,), float("nan"), device="cuda")
            eng.score_dev(list(natives), d_in.data_ptr(), n, L, lut, None, d_mean.data_ptr())     # the engine's own planes
            eng.sync()
            assert np.array_equal(d_mean.cpu().numpy().view(np.uint32), want.view(np.uint32)), fuse
            _, host_mean = eng.score(list(natives), b, lut, want_matrix=False, want_mean=True)     # host call, mean only
            assert np.array_equal(host_mean.view(np.uint32), want.view(np.uint32)), fuse
        # a character outside the alphabet is still reported, and the launch after it is clean
        bad = b.copy(); bad[n // 3, L - 1] = ord("!")
        d_bad = torch.from_numpy(bad).cuda()
        eng.score_mean_planes_dev(list(natives), d_bad.data_ptr(), n, L, lut, planes.data_ptr(), stride, d_mean.data_ptr())
        with pytest.raises(ValueError):
            eng.sync()
        d_mean.fill_(float("nan"))
        eng.score_mean_planes_dev(list(natives), d_in.data_ptr(), n, L, lut, planes.data_ptr(), stride, d_mean.data_ptr())
        eng.sync()
        assert np.array_equal(d_mean.cpu().numpy().view(np.uint32), want.view(np.uint32))
    finally:
        ab_option(eng, "fuse_mean_batch", 0)


@pytest.mark.parametrize("L,H,M,n", [(8, 200, 1, 100_000), (8, 200, 1, 65_536 + 16 * 3), (8, 256, 2, 70_001), (14, 200, 3, 30_000), (8, 130, 1, 66_000),
                                     (8, 200, 1, 4_000)])
def test_wide_cnn_head_through_lds_slabs_gives_the_same_bits(eng, L, H, M, n):
    """Round 6: a 4-letter CNN with more than 128 hidden units at batch size runs as the conv-only kernel + a head kernel that streams the
    H x H layer through LDS slabs once per lockstep round of 8 tiles (`cnn_head_slab`, k_cnn_head_slab) instead of the fused kernel, whose
    waves stream it from L2 per tile.  Same MFMA sequence per output element: the SAME BITS as the fused kernel (`cnn_head_slab` = 0);
    both agree with the oracle at the head and the tail of the batch (ragged last tile, a last round of one to three tiles walked
    without slabs); small batches keep the fused kernel."""
    alpha = "TGCA" if L == 8 else "UGCA"
    natives, ws = zip(*[make_native(eng, "cnn", L, 4, H, 32, 5, seed=700 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, seqs = rand_seqs(n, L, alpha, seed=n % 83 + H)
    try:
        outs = {}
        for slab in (0, 1):
            eng.set_option("cnn_head_slab", slab)
            outs[slab], mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[slab], axis=1))
        assert np.array_equal(outs[1].view(np.uint32), outs[0].view(np.uint32))
        k = min(n, 300)
        for m in range(M):
            assert_scores(outs[1][:k, m], ref_np.keras_fitness(seqs[:k], alpha, "cnn", ws[m], exact=True), f"cnn L={L} H={H} slab head member {m}")
            assert_scores(outs[1][n - k:, m], ref_np.keras_fitness(seqs[n - k:], alpha, "cnn", ws[m], exact=True), f"cnn L={L} H={H} slab head member {m} tail")
        bad = b.copy(); bad[n - 2, L - 1] = ord("!")
        with pytest.raises(ValueError):
            eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[1])
    finally:
        eng.set_option("cnn_head_slab", 1)


@pytest.mark.parametrize("L,H,K,n", [(20, 200, 5, 70_000), (12, 256, 3, 66_001)])
def test_wide_protein_cnn_head_through_lds_slabs(eng, L, H, K, n):
    """The 20-letter CNN with more than 128 hidden units always runs conv kernel + head kernel; at batch size the head is the slab form too
    (k_cnn_head_slab): same bits as the head that streams from L2 (`cnn_head_slab` = 0), both held to the oracle."""
    nm, w = make_native(eng, "cnn", L, 20, H, 32, K, seed=990)
    lut = _native.make_lut(s_utils.AAS)
    b, _ = rand_seqs(n, L, s_utils.AAS, seed=L + H)
    try:
        outs = {}
        for slab in (0, 1):
            eng.set_option("cnn_head_slab", slab)
            outs[slab], _ = eng.score([nm], b, lut)
        assert np.array_equal(outs[1].view(np.uint32), outs[0].view(np.uint32))
        k = 200
        assert_scores(outs[1][:k, 0], c_oracle.forward("cnn", lut[b[:k]], 20, w), f"protein cnn L={L} H={H} slab head")
        assert_scores(outs[1][n - k:, 0], c_oracle.forward("cnn", lut[b[n - k:]], 20, w), f"protein cnn L={L} H={H} slab head tail")
    finally:
        eng.set_option("cnn_head_slab", 1)


@pytest.mark.parametrize("L,alpha,H,M,n", [(90, s_utils.AAS, 200, 1, 100_000), (90, s_utils.AAS, 100, 2, 40_003), (33, s_utils.AAS, 200, 1, 70_000),
                                           (237, s_utils.AAS, 100, 1, 20_000), (400, "UGCA", 100, 1, 30_000), (31, s_utils.AAS, 256, 1, 66_000),
                                           (90, s_utils.AAS, 64, 3, 25_000), (90, s_utils.AAS, 200, 1, 2_000), (30, s_utils.AAS, 200, 1, 700_001)])
def test_mlp_first_layer_position_major_gives_the_same_bits(eng, L, alpha, H, M, n):
    """Round 6: an MLP whose first-layer rows do not fit LDS (protein alphabets: dyna_ppo.py:54's MLP(seq_len, 200, alphabet) on AAV is
    90 x 20 rows of 200 floats) gathered seq_len rows PER SEQUENCE from L2 -- 0.03-0.27 of the MFMA rate.  At batch size the first layer is
    now taken position-major by its own kernel (`mlp_l1_pos`, k_mlp_l1_pos: a position's rows cross L2 -> LDS once per 16-32 tiles) and the
    dense kernel starts from its scratch.  Bias + rows in position order either way: the SAME BITS as the gather form (`mlp_l1_pos` = 0),
    both held to the oracle at the head and the tail of the batch (ragged last tile, odd lengths, several members); a character outside
    the alphabet is still reported; small batches keep the gather form; a batch whose scratch would pass 512 MB goes in slices (the 7e5 case)."""
    A = len(alpha)
    natives, ws = zip(*[make_native(eng, "mlp", L, A, H, seed=800 + m) for m in range(M)])
    lut = _native.make_lut(alpha)
    b, _ = rand_seqs(n, L, alpha, seed=L + H + n % 7)
    try:
        outs = {}
        for pos in (0, 1):
            eng.set_option("mlp_l1_pos", pos)
            outs[pos], mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[pos], axis=1))
        assert np.array_equal(outs[1].view(np.uint32), outs[0].view(np.uint32))
        k = min(n, 200)
        for m in range(M):
            assert_scores(outs[1][:k, m], c_oracle.forward("mlp", lut[b[:k]], A, ws[m]), f"mlp L={L} A={A} H={H} position-major member {m}")
            assert_scores(outs[1][n - k:, m], c_oracle.forward("mlp", lut[b[n - k:]], A, ws[m]), f"mlp L={L} A={A} H={H} position-major member {m} tail")
        for where, col in ((n - 1, L - 1), (n // 2, 0), (17, L // 2)):
            bad = b.copy(); bad[where, col] = ord("!")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[1])
    finally:
        eng.set_option("mlp_l1_pos", 1)


@pytest.mark.parametrize("L,M,n", [(40, 3, 1000), (40, 3, 2000), (33, 1, 3000), (40, 8, 500), (90, 3, 1990), (40, 3, 1370), (25, 2, 4001)])
def test_protein_cnn_mid_size_calls_are_segmented(eng, L, M, n):
    """Round 6: the protein CNN between half a unit and two units per CU.  A unit (member, 16-sequence tile) is a serial walk over the
    positions by one wave pair -- 1.47 ms at 237 residues -- and up to one unit per CU the four pairs of a workgroup now split its tile's
    positions (`cnn_pair_seg` < 0, SB = 1: 3 x 1000 GFP sequences 1.5 ms -> 0.47 ms); between one and two units per CU the call is two
    such launches of half the tiles.  Every conv3 output sees the same MFMA sequence as in the whole-sequence walk: the SAME BITS as
    `cnn_pair_seg` = 0, both held to the oracle; ragged last tiles, a bad character in either half."""
    natives, ws = zip(*[make_native(eng, "cnn", L, 20, 100, 32, 5, seed=900 + m) for m in range(M)])
    lut = _native.make_lut(s_utils.AAS)
    b, _ = rand_seqs(n, L, s_utils.AAS, seed=L + M + n)
    try:
        outs = {}
        for seg in (0, -1):
            eng.set_option("cnn_pair_seg", seg)
            outs[seg], mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
            assert np.array_equal(mean, np.mean(outs[seg], axis=1))
        assert np.array_equal(outs[-1].view(np.uint32), outs[0].view(np.uint32))
        k = min(n, 150)
        for m in range(M):
            assert_scores(outs[-1][:k, m], c_oracle.forward("cnn", lut[b[:k]], 20, ws[m]), f"protein cnn mid L={L} member {m}")
            assert_scores(outs[-1][n - k:, m], c_oracle.forward("cnn", lut[b[n - k:]], 20, ws[m]), f"protein cnn mid L={L} member {m} tail")
        for where in (3, n - 1, n // 2 + 17):
            bad = b.copy(); bad[where, L // 2] = ord("B")
            with pytest.raises(ValueError):
                eng.score(list(natives), bad, lut)
        again, _ = eng.score(list(natives), b, lut)
        assert np.array_equal(again, outs[-1])
    finally:
        eng.set_option("cnn_pair_seg", -1)
