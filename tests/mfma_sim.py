"""Lane-level NumPy model of the MFMA scoring kernels (test infrastructure).

Re-executes flexs_amd/csrc/score_cnn_mfma.hip / score_dense_mfma.hip step by
step for ONE tile of 16 sequences, with the 64 lanes of a wave as array axis 0
and `v_mfma_f32_16x16x4_f32` modelled from its documented operand layout:

    A[i][k] = a[lane = 16*k + i]      B[k][j] = b[lane = 16*k + j]
    D[4*g + r][j] lives in lane 16*g + j, register r

It consumes the REAL packed weight buffer produced by the library
(fx_debug_pack_weights), so it checks -- on a machine without a GPU -- that the
host-side fragment packing, the k-step <-> channel mapping, the sliding-window
schedule and the final cross-lane reduction compute the reference network.
Arithmetic is float64 so the comparison with the oracle is tight.
"""
import numpy as np

LANES = np.arange(64)
G = LANES >> 4
SQ = LANES & 15


def mfma16(a, b, c):
    """a, b: (64,) per-lane scalars; c: (64, 4).  Returns c + A*B in the C/D layout."""
    A = a.reshape(4, 16).T            # A[i, k]
    B = b.reshape(4, 16)              # B[k, j]
    D = A @ B                         # D[row, col]
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, SQ]
    return out


def rl_of(lay):
    """k-steps of the last hidden tile the kernels run (4 when H was rounded up to a larger tile count)."""
    return lay["RLH"] if lay["HTR"] == lay["HT"] else 4


def blocks(packed, off, idx):
    """f4 block `idx` starting at float offset `off`: (64 lanes, 4)."""
    return packed[off + idx * 256: off + (idx + 1) * 256].reshape(64, 4)


def mma_layer(packed, off, TI, TO, inp, acc, rl_last=4):
    """inp: list TI of (64,4); acc: list TO of (64,4) -- mirrors mma_layer<> in mfma_common.h"""
    for mi in range(TI):
        a = [blocks(packed, off, mi * TO + mo) for mo in range(TO)]
        for r in range(4):
            if mi == TI - 1 and r >= rl_last:
                break
            for mo in range(TO):
                acc[mo] = mfma16(a[mo][:, r], inp[mi][:, r], acc[mo])
    return acc


def init_bias(packed, off, TO):
    return [np.stack([packed[off + 16 * mo + 4 * G + r] for r in range(4)], axis=1).astype(np.float64)
            for mo in range(TO)]


def final_dot(packed, off_w, bout, h, HT):
    out = np.zeros(64)
    for mo in range(HT):
        for r in range(4):
            out += packed[off_w + 16 * mo + 4 * G + r] * h[mo][:, r]
    # __shfl_xor 16 then 32
    out = out + out[LANES ^ 16]
    out = out + out[LANES ^ 32]
    return out + bout


def cnn_tile(packed, lay, codes16, A, K, F, H, conv1_gather=True):
    """codes16: (16, L) alphabet indices of the tile's sequences -> (16,) scores."""
    packed = packed.astype(np.float64)
    L = codes16.shape[1]
    L1 = L - K + 1
    K3 = A - 1
    FT, HT = lay["FT"], lay["HT"]
    PL2 = (K - 1) // 2
    PR2 = K - 1 - PL2
    PL3 = (K3 - 1) // 2
    PR3 = K3 - 1 - PL3
    S1 = (K * A + 3) // 4
    zero = lambda T: [np.zeros((64, 4)) for _ in range(T)]  # noqa: E731
    win1 = [zero(FT) for _ in range(K)]
    win2 = [zero(FT) for _ in range(K3)]
    gmax = zero(FT)
    code = codes16[SQ].astype(np.int64)                  # (64, L): lane's own sequence
    for s in range(L1 + PR2 + PR3):
        win1 = win1[1:] + [None]
        win2 = win2[1:] + [None]
        if s < L1:
            o1 = init_bias(packed, lay["off_cb"], FT)
            if conv1_gather:
                for j in range(K):
                    rowp = lay["off_w1p"] + (j * A + code[:, s + j]) * (16 * FT + 4) + 4 * G     # (rows FX_C1_ROW(FT) floats apart)
                    for mo in range(FT):
                        o1[mo] = o1[mo] + np.stack([packed[rowp + 16 * mo + r] for r in range(4)], axis=1)
            else:
                for st in range(S1):
                    j, a0 = (4 * st) // A, (4 * st) % A
                    sg, r = st >> 2, st & 3
                    b = (code[:, s + j] == a0 + G).astype(np.float64)
                    for mo in range(FT):
                        a = blocks(packed, lay["off_first"], sg * FT + mo)[:, r]
                        o1[mo] = mfma16(a, b, o1[mo])
            win1[K - 1] = [np.maximum(x, 0) for x in o1]
        else:
            win1[K - 1] = zero(FT)
        t2 = s - PR2
        if 0 <= t2 < L1:
            o2 = init_bias(packed, lay["off_cb"] + 16 * FT, FT)
            for j in range(K):
                if 0 <= t2 + j - PL2 < L1:
                    o2 = mma_layer(packed, lay["off_c2"] + j * FT * FT * 256, FT, FT, win1[j], o2)
            win2[K3 - 1] = [np.maximum(x, 0) for x in o2]
        else:
            win2[K3 - 1] = zero(FT)
        t3 = t2 - PR3
        if 0 <= t3 < L1:
            o3 = init_bias(packed, lay["off_cb"] + 32 * FT, FT)
            for j in range(K3):
                if 0 <= t3 + j - PL3 < L1:
                    o3 = mma_layer(packed, lay["off_c3"] + j * FT * FT * 256, FT, FT, win2[j], o3)
            gmax = [np.maximum(g, x) for g, x in zip(gmax, o3)]
    db = lay["off_db"]
    h1 = init_bias(packed, db, HT)
    h1 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d1"], FT, HT, gmax, h1)]
    h2 = init_bias(packed, db + 16 * HT, HT)
    h2 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d2"], HT, HT, h1, h2, rl_of(lay))]
    y = final_dot(packed, db + 32 * HT, packed[db + 48 * HT], h2, HT)
    return y[:16]                                        # lanes of group 0 hold the 16 sequences


def mlp_tile(packed, lay, codes16, A, H, l1_gather=True):
    """MLP: one-hot first layer on MFMA (k index = l*A + a), two HxH layers, dot."""
    packed = packed.astype(np.float64)
    L = codes16.shape[1]
    HT = lay["HT"]
    code = codes16[SQ].astype(np.int64)
    S1 = (L * A + 3) // 4
    db = lay["off_db"]
    h = init_bias(packed, db, HT)
    if l1_gather:
        for l in range(L):
            rowp = lay["off_w1p"] + (l * A + code[:, l]) * (16 * HT) + 4 * G
            for mo in range(HT):
                h[mo] = h[mo] + np.stack([packed[rowp + 16 * mo + r] for r in range(4)], axis=1)
    else:
        for st in range(S1):
            sg, r = st >> 2, st & 3
            k = 4 * st + G                                   # flattened feature index of this lane group
            l, a = k // A, k % A
            valid = k < L * A
            b = np.where(valid, (code[LANES, np.minimum(l, L - 1)] == a), False).astype(np.float64)
            for mo in range(HT):
                aw = blocks(packed, lay["off_first"], sg * HT + mo)[:, r]
                h[mo] = mfma16(aw, b, h[mo])
    h = [np.maximum(x, 0) for x in h]
    h2 = init_bias(packed, db + 16 * HT, HT)
    h2 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d2"], HT, HT, h, h2, rl_of(lay))]
    h3 = init_bias(packed, db + 32 * HT, HT)
    h3 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d3"], HT, HT, h2, h3, rl_of(lay))]
    y = final_dot(packed, db + 48 * HT, packed[db + 64 * HT], h3, HT)
    return y[:16]


def ge_tile(packed, lay, codes16, A, H):
    """GlobalEpistasis: scalar gather-sum, 1 -> H on the VALU in B-operand layout, HxH MFMA, dot."""
    packed = packed.astype(np.float64)
    L = codes16.shape[1]
    HT = lay["HT"]
    code = codes16[SQ].astype(np.int64)
    db = lay["off_db"]
    s = np.zeros(64)
    for l in range(L):                                   # lane group g takes positions l = g (mod 4)
        s = s + np.where(l % 4 == G, packed[lay["off_first"] + l * A + code[:, l]], 0.0)
    s = s + s[LANES ^ 16]
    s = s + s[LANES ^ 32]
    s = np.maximum(s + packed[db], 0)
    w2 = init_bias(packed, db + 4, HT)
    b2 = init_bias(packed, db + 4 + 16 * HT, HT)
    h = [np.maximum(b2[mo] + s[:, None] * w2[mo], 0) for mo in range(HT)]
    h2 = init_bias(packed, db + 4 + 32 * HT, HT)
    h2 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d3"], HT, HT, h, h2, rl_of(lay))]
    y = final_dot(packed, db + 4 + 48 * HT, packed[db + 4 + 64 * HT], h2, HT)
    return y[:16]


def cnn_pair_tile(packed, lay, codes16, A, K, F, H):
    """score_cnn_pair.hip: two waves per tile, wave `mo` owns output-channel tile `mo` of conv2 /
    conv3; conv3 in scatter form with a window of A-1 partial sums; halves swapped through LDS."""
    packed = packed.astype(np.float64)
    L = codes16.shape[1]
    L1 = L - K + 1
    K3 = A - 1
    FT, HT = lay["FT"], lay["HT"]
    assert FT == 2
    PL2 = (K - 1) // 2
    PR2 = K - 1 - PL2
    PL3 = (K3 - 1) // 2
    PR3 = K3 - 1 - PL3
    code = codes16[SQ].astype(np.int64)
    cb = lay["off_cb"]

    def bias_tile(off, mo):
        return np.stack([packed[off + 16 * mo + 4 * G + r] for r in range(4)], axis=1)

    # per-wave state (index = mo)
    win1 = [[[np.zeros((64, 4)) for _ in range(FT)] for _ in range(K)] for _ in range(2)]
    accw = [[bias_tile(cb + 32 * FT, mo) for _ in range(K3)] for mo in range(2)]
    gmax = [np.zeros((64, 4)) for _ in range(2)]
    for s in range(L1 + PR2 + PR3):
        t2 = s - PR2
        mine = [None, None]
        for mo in range(2):
            win1[mo] = win1[mo][1:] + [None]
            if s < L1:
                o1 = [bias_tile(cb, t) for t in range(FT)]
                for j in range(K):
                    rowp = lay["off_w1p"] + (j * A + code[:, s + j]) * (16 * FT + 4) + 4 * G     # (rows FX_C1_ROW(FT) floats apart)
                    for t in range(FT):
                        o1[t] = o1[t] + np.stack([packed[rowp + 16 * t + r] for r in range(4)], axis=1)
                win1[mo][K - 1] = [np.maximum(x, 0) for x in o1]
            else:
                win1[mo][K - 1] = [np.zeros((64, 4)) for _ in range(FT)]
            if 0 <= t2 < L1:
                o2a = bias_tile(cb + 16 * FT, mo)
                o2b = np.zeros((64, 4))
                for j in range(K):
                    a0 = blocks(packed, lay["off_c2"], (j * FT + 0) * FT + mo)
                    a1 = blocks(packed, lay["off_c2"], (j * FT + 1) * FT + mo)
                    for r in range(4):
                        o2a = mfma16(a0[:, r], win1[mo][j][0][:, r], o2a)
                        o2b = mfma16(a1[:, r], win1[mo][j][1][:, r], o2b)
                mine[mo] = np.maximum(o2a + o2b, 0)
        for mo in range(2):
            if 0 <= t2 < L1:
                out2 = [mine[0], mine[1]]                # after the LDS swap both waves hold both halves
                for j in range(K3):
                    for mi in range(FT):
                        a = blocks(packed, lay["off_c3"], (j * FT + mi) * FT + mo)
                        for r in range(4):
                            accw[mo][K3 - 1 - j] = mfma16(a[:, r], out2[mi][:, r], accw[mo][K3 - 1 - j])
            t3f = t2 - PR3
            if 0 <= t3f < L1:
                gmax[mo] = np.maximum(gmax[mo], accw[mo][0])
            accw[mo] = accw[mo][1:] + [bias_tile(cb + 32 * FT, mo)]
    db = lay["off_db"]
    pooled = [gmax[0], gmax[1]]
    h1 = init_bias(packed, db, HT)
    h1 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d1"], FT, HT, pooled, h1)]
    h2 = init_bias(packed, db + 16 * HT, HT)
    h2 = [np.maximum(x, 0) for x in mma_layer(packed, lay["off_d2"], HT, HT, h1, h2, rl_of(lay))]
    return final_dot(packed, db + 32 * HT, packed[db + 48 * HT], h2, HT)[:16]
