// Part of the training step (train_core.h includes it; not a stand-alone header): the one contraction routine fxt_gemm, the address-space-qualified pointer types, and its staged / F = 32 conv forms.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// C[m][n] = sum over (ko, ki) of A(m, ko, ki) * B(ko, ki, n),  m < Md, n < Nd, ko < Ko, ki < Ki.
// The contraction index is kept as a PAIR so that conv taps / batch rows never need a division in the inner loop;
// each ko runs ceil(Ki / 4) k-steps (the overhang multiplies zeros).  FA: prep(m, kq) -> per-lane state (once per
// tile), at(state, ko, k0) = A(m, ko, k0 + kq); FB: prep(n, kq), at(state, ko, k0) = B(ko, k0 + kq, n); FC: put(m, n, value).
// `first_wave`: the wave that takes tile 0 (tiles go round-robin from there).  Two products of one phase -- the few long tiles of
// an input gradient and the many short ones of a weight gradient -- are dealt as ONE sequence: the second call passes
// fxt_tiles(first product) as its first_wave, so no wave gets a long tile on top of a full share of short ones.
FXT_HD int fxt_tiles(int Md, int Nd) { return ((Md + 15) >> 4) * ((Nd + 15) >> 4); }
// `split` (device, LDS scratch of FXT_SPLIT_FLOATS floats, or null): a product with FEW tiles and a LONG contraction -- conv2
// forward: 4 tiles of 40 k-steps for 16 waves -- is cut along the contraction as well: the groups of eight k-steps of a tile are
// dealt to up to four waves, the partial tiles meet in the scratch (one LDS-only barrier), and the first wave of a tile adds them
// in split order and runs the epilogue.  How a product is cut depends on its shape and the workgroup size only
// (fxt_split_ways), so every instantiation of this source -- shape-agnostic or canonical -- sums in the same order.
// ALL waves of the workgroup must call fxt_gemm together when `split` is given (the barrier).
#define FXT_SPLIT_FLOATS 4096
FXT_HD int fxt_split_ways(int tiles, int Ko, int Ki, int nw) {
    if (Ki < 32 || tiles * 2 > nw) return 1;
    const int groups = Ko * ((Ki + 31) >> 5);
    int ways = nw / tiles;
    if (ways > 4) ways = 4;
    if (ways > groups) ways = groups;
    return ways < 1 ? 1 : ways;
}
// workgroup jobs of a product (tiles x the ways it is cut): what the NEXT product of the phase passes as its first_wave
FXT_HD int fxt_jobs(int Md, int Nd, int Ko, int Ki, int nw, bool can_split) {
    const int t = fxt_tiles(Md, Nd);
#if defined(FX_AB)
    return t * (can_split ? fxt_split_ways(t, Ko, Ki, nw) : 1);
#else
    (void)Ko; (void)Ki; (void)nw; (void)can_split;
    return t;
#endif
}
template <class FA, class FB, class FC, class SP = float*>
FXT_HD void fxt_gemm(const FxtWg& wg, int Md, int Nd, int Ko, int Ki, const FA& fa, const FB& fb, const FC& fc, int first_wave = 0,
                     SP split = nullptr) {
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8;                     // k-steps whose operand loads are in flight together
    const int lane = wg.tid & 63, nw = wg.nthr >> 6;
    const int wave = ((wg.tid >> 6) + nw - first_wave % nw) % nw;
    const int i = lane & 15, kq = lane >> 4;
    const int tn = (Nd + 15) >> 4, tiles = ((Md + 15) >> 4) * tn;
    const int T = Ko * ((Ki + 3) >> 2);      // k-steps of the whole contraction, (ko, k0) in row-major order
    // (measured: the cut LOSES -- 24.6 -> 33.1 us per forward+backward launch of the 3 x CNN step, every phase it touches
    //  slower: the extra barrier and the partial tiles through LDS cost more than the idle waves were worth,
    //  profiles/r4_train_split_ab.log -- so it is compiled into the A/B build only; elsewhere ways == 1 folds it all away)
#if defined(FX_AB)
    const int ways = split ? fxt_split_ways(tiles, Ko, Ki, nw) : 1;
#else
    constexpr int ways = 1;
    (void)split;
#endif
    const int gpk = (Ki + 31) >> 5, groups = Ko * gpk;
    auto epilogue = [&](int t, f4_t acc) {
        const int m0 = (t / tn) << 4, n = ((t % tn) << 4) + i;
        if (n < Nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = m0 + 4 * kq + r;
                if (mr < Md) fc.put(mr, n, acc[r]);
            }
        }
    };
    for (int job = wave; job < tiles * ways; job += nw) {
        const int t = ways > 1 ? job / ways : job, sp = ways > 1 ? job - t * ways : 0;
        const int g_lo = groups * sp / ways, g_hi = groups * (sp + 1) / ways;      // this job's groups of eight k-steps (ways > 1)
        const int m0 = (t / tn) << 4, n0 = (t % tn) << 4;
        const int m = m0 + i, n = n0 + i;
        const bool mok = m < Md, nok = n < Nd;
        const auto sa = fa.prep(mok ? m : 0, kq);
        const auto sb = fb.prep(nok ? n : 0, kq);
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        // The operands come from L2 / LDS through index functors: issued one k-step at a time every MFMA would wait a
        // full memory round trip (the first build ran at ~1 us per k-step).  U k-steps are loaded first, then multiplied.
        // Rows past Md / columns past Nd need no masking: row i of A only reaches row i of the product, column j of B only
        // column j, and those are never stored (their lanes read row / column 0).  Only the contraction index must be
        // exact: a k-step past Ki has to contribute zero.
        if (Ki >= 4 * U) {
            // long inner index (conv taps x channels, dense layers): whole groups of U k-steps inside one `ko` need no
            // range logic at all -- (ko, k0) are wave-uniform, the U fetches differ by constant offsets; the functors'
            // own checks (conv positions) depend on ko only
            for (int ko = 0; ko < Ko; ++ko) {
                int k0 = 0;
                for (; k0 + 4 * U <= Ki; k0 += 4 * U) {
                    if (ways > 1) { const int gi = ko * gpk + (k0 >> 5); if (gi < g_lo || gi >= g_hi) continue; }
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { a[u] = fa.at(sa, ko, k0 + 4 * u); b[u] = fb.at(sb, ko, k0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
                }
                if (k0 < Ki && (ways == 1 || (ko * gpk + (k0 >> 5) >= g_lo && ko * gpk + (k0 >> 5) < g_hi))) {   // the row's last, partial group
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int kk = k0 + 4 * u;
                        const bool live = kk < Ki;           // wave-uniform
                        const bool kok = kk + kq < Ki;
#if defined(FXT_EMUL)     // (the emulator does not perform a load whose value is masked away: on the device it reads -- and drops -- a
                          //  neighbouring array's element, which a race detector reports and an exact-size buffer cannot hold)
                        a[u] = kok ? fa.at(sa, ko, kk) : 0.f;
                        b[u] = kok ? fb.at(sb, ko, kk) : 0.f;
#else
                        const float av = fa.at(sa, ko, live ? kk : 0), bv = fb.at(sb, ko, live ? kk : 0);
                        a[u] = kok ? av : 0.f;
                        b[u] = kok ? bv : 0.f;
#endif
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (k0 + 4 * u < Ki) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
                }
            }
        } else if (Ki <= 4) {
            // one k-step per `ko` (conv weight gradients of short sequences: the four positions of a row): k-step = ko
            for (int s = 0; s < Ko; s += U) {
                float a[U], b[U];
                const bool kin = kq < Ki;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = s + u < Ko;            // wave-uniform
#if defined(FXT_EMUL)
                    a[u] = (live && kin) ? fa.at(sa, s + u, 0) : 0.f;
                    b[u] = (live && kin) ? fb.at(sb, s + u, 0) : 0.f;
#else
                    const float av = fa.at(sa, live ? s + u : 0, 0), bv = fb.at(sb, live ? s + u : 0, 0);
                    a[u] = (live && kin) ? av : 0.f;
                    b[u] = (live && kin) ? bv : 0.f;
#endif
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s + u < Ko) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
        } else {
            // short inner index (weight gradients: positions of a row, rows of a slice): the (ko, k0) pairs are walked as
            // one flat sequence of k-steps, wave-uniform counters instead of a division
            int ko = 0, k0 = 0;
            for (int s = 0; s < T; s += U) {
                float a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = s + u < T;
                    const bool kok = live && k0 + kq < Ki;
#if defined(FXT_EMUL)
                    a[u] = kok ? fa.at(sa, ko, k0) : 0.f;
                    b[u] = kok ? fb.at(sb, ko, k0) : 0.f;
#else
                    const float av = fa.at(sa, live ? ko : 0, live ? k0 : 0);
                    const float bv = fb.at(sb, live ? ko : 0, live ? k0 : 0);
                    a[u] = kok ? av : 0.f;
                    b[u] = kok ? bv : 0.f;
#endif
                    k0 += 4;
                    if (k0 >= Ki) { k0 = 0; ++ko; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s + u < T) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
        }
        if (ways > 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) split[(job * 64 + lane) * 4 + r] = acc[r];
        } else {
            epilogue(t, acc);
        }
    }
    if (ways > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        for (int t = wave; t < tiles; t += nw) {
            f4_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = split[((t * ways) * 64 + lane) * 4 + r];
            for (int sp = 1; sp < ways; ++sp)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += split[((t * ways + sp) * 64 + lane) * 4 + r];
            epilogue(t, acc);
        }
    }
#else
    (void)wg; (void)first_wave; (void)split;
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < Nd; ++n) {
            float acc = 0.f;
            for (int ko = 0; ko < Ko; ++ko)
                for (int ki = 0; ki < Ki; ++ki)          // element ki belongs to lane group kq = ki % 4 of k-step k0 = ki - kq
                    acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko, ki & ~3), acc);
            fc.put(m, n, acc);
        }
#endif
}

// ---- address spaces ---------------------------------------------------------------------------------------------
// A pointer whose address space the compiler does not know is read with flat_load, and a FLAT access that resolves to
// LDS is several times slower than ds_read (phase timeline, profiles/r3_train_trace.log: ~1.1 us per group of eight
// k-steps with every operand in LDS).  The step is therefore compiled per placement -- workspace in LDS (3) or global
// memory (1), weights in LDS or global memory -- with address-space-qualified pointer types; the host build has one.
#if defined(FXT_EMUL)
typedef float fxt_f4 __attribute__((ext_vector_type(4), aligned(4)));   // (host memory of the emulator: no 16-byte promise)
template <int AS> struct FxtMem { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const fxt_f4* CF4; typedef fxt_f4* F4; };
#elif FXT_DEVICE
typedef float fxt_f4 __attribute__((ext_vector_type(4)));
template <int AS> struct FxtMem {
    typedef __attribute__((address_space(AS))) float* F;
    typedef const __attribute__((address_space(AS))) float* CF;
    typedef __attribute__((address_space(AS))) int* I;
    typedef const __attribute__((address_space(AS))) int* CI;
    typedef const __attribute__((address_space(AS))) fxt_f4* CF4;      // sixteen bytes at a time (fxt_gemm_staged)
    typedef __attribute__((address_space(AS))) fxt_f4* F4;
};
template <> struct FxtMem<0> { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const fxt_f4* CF4; typedef fxt_f4* F4; };   // (flat / host)
#else
template <int AS> struct FxtMem { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const float* CF4; };   // (CF4: never dereferenced by the host build)
#endif


// fxt_gemm for a conv product whose B operand -- taps [0, Ko) of a conv kernel, `rows_per_tap` rows of F floats each in global
// memory at `wsrc` -- is staged through `wbuf` (LDS on the device; rows `ldw` floats apart) in groups of G taps.  `fb` is the B functor
// built OVER wbuf: it is handed the tap index relative to its group.  Ki is a multiple of 32 (whole groups of eight k-steps), a wave
// owns at most FXT_STAGED_TPW tiles whose accumulators stay in registers across the groups (the caller checks both: fxt_staged_ok).
// ALL threads of the workgroup call it together (two LDS barriers per group).
#define FXT_STAGED_TPW 2
FXT_HD bool fxt_staged_ok(int Md, int Nd, int Ki, int F, int nw) { return (Ki & 31) == 0 && (F & 3) == 0 && fxt_tiles(Md, Nd) <= nw * FXT_STAGED_TPW; }
template <int WSAS, int WAS, class FA, class FB, class FC>
FXT_HD void fxt_gemm_staged(const FxtWg& wg, int Md, int Nd, int Ko, int Ki, const FA& fa, const FB& fb, const FC& fc,
                            typename FxtMem<WAS>::CF wsrc, typename FxtMem<WSAS>::F wbuf, int G, int rows_per_tap, int F, int ldw) {
    if (G < 1) G = Ko;                                     // (never from the host's sizing; a group must advance)
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8, TPW = FXT_STAGED_TPW;
    const int lane = wg.tid & 63, nw = wg.nthr >> 6, wave = wg.tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int tn = (Nd + 15) >> 4, tiles = ((Md + 15) >> 4) * tn;
    f4_t acc[TPW];
    decltype(fa.prep(0, 0)) sa[TPW];
    decltype(fb.prep(0, 0)) sb[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * nw;
        const int m = ((t / tn) << 4) + i, n = ((t % tn) << 4) + i;
        acc[q] = f4_t{0.f, 0.f, 0.f, 0.f};
        sa[q] = fa.prep((t < tiles && m < Md) ? m : 0, kq);
        sb[q] = fb.prep((t < tiles && n < Nd) ? n : 0, kq);
    }
    const int f4_per_row = F >> 2;
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        fxt_sync_ws<WSAS>();                               // everybody is through with the previous group's taps (or the previous phase)
        const int pieces = (g1 - g0) * rows_per_tap * f4_per_row;
        for (int p = wg.tid; p < pieces; p += wg.nthr) {
            const int row = p / f4_per_row, c4 = p - row * f4_per_row;
            const f4_t v = *(typename FxtMem<WAS>::CF4)(wsrc + ((g0 * rows_per_tap + row) * F + 4 * c4));
            const int o = row * ldw + 4 * c4;               // (scalar stores: they keep wbuf's address space -- ds_write, not flat)
            wbuf[o] = v[0]; wbuf[o + 1] = v[1]; wbuf[o + 2] = v[2]; wbuf[o + 3] = v[3];
        }
        fxt_sync_ws<WSAS>();
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (wave + q * nw >= tiles) continue;          // (wave-uniform)
            for (int ko = g0; ko < g1; ++ko) {
                for (int k0 = 0; k0 < Ki; k0 += 4 * U) {
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { a[u] = fa.at(sa[q], ko, k0 + 4 * u); b[u] = fb.at(sb[q], ko - g0, k0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[q], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * nw;
        if (t >= tiles) continue;
        const int m0 = (t / tn) << 4, n = ((t % tn) << 4) + i;
        if (n < Nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = m0 + 4 * kq + r;
                if (mr < Md) fc.put(mr, n, acc[q][r]);
            }
        }
    }
#else
    (void)wg;
    float* accs = new float[(size_t)Md * Nd]();
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        for (int row = 0; row < (g1 - g0) * rows_per_tap; ++row)
            for (int c = 0; c < F; ++c) wbuf[row * ldw + c] = wsrc[(g0 * rows_per_tap + row) * F + c];
        for (int m = 0; m < Md; ++m)
            for (int n = 0; n < Nd; ++n) {
                float acc = accs[(size_t)m * Nd + n];
                for (int ko = g0; ko < g1; ++ko)
                    for (int ki = 0; ki < Ki; ++ki)
                        acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko - g0, ki & ~3), acc);
                accs[(size_t)m * Nd + n] = acc;
            }
    }
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < Nd; ++n) fc.put(m, n, accs[(size_t)m * Nd + n]);
    delete[] accs;
#endif
}

// ---- MODE 3 (round 5): the conv products of the long protein CNNs, F = 32 filters ---------------------------------------------
// Measured in round 5's first session (profiles/r5_train_gfp_*): the staged form (MODE 2) runs conv3 forward at 0.36 and its
// backward at 0.30 of the f32-MFMA rate -- one A and one B ds_read (the B one 2-way conflicted: rows F + 4 apart) and ~10 address
// instructions per MFMA, the staging copy's L2 round trip exposed between two barriers per tap group, and the weight gradient
// (78 tiles x 59 k-steps) fetching both operands per MFMA through the heaviest index functors.  Three changes, same bits:
//   * fxt_conv32_staged: a wave owns a 16-row M tile and BOTH 16-column N tiles (F = 32): one A fetch feeds two MFMAs.  The tap
//     group's kernels are kept in LDS with 32-float rows, rotated so that the fetch is conflict-free WITHOUT padding -- forward:
//     element (c, n) at column (n + 16 c) mod 32, transposed read of the input gradient: element (c, o) at column (o + 2 c) mod 32
//     (a product stages its own copy, so each picks its rotation) -- seven taps per group instead of six in the same bytes; the
//     next group's rows are fetched from L2 into registers BEFORE the current group's MFMAs and stored behind them;
//   * fxt_conv32_wgrad: dW[j][c][o] = sum_t x[t + j - pl][c] dz[t][o].  Taps j and j + 4 read the same x k-step blocks one k-step
//     apart, so a wave owns taps {res, res + 4, ...} x 16 channels x 16 out channels (16 jobs = 4 residues x 2 x 2): per k-step ONE
//     new x block and ONE dz block from LDS feed up to five MFMAs out of a register window.  The bias row rides on the residue-3
//     waves (four taps of conv3's nineteen).
// Every output element still sums the same products in the same order through the same instruction: the SAME BITS as fxt_gemm.
// The tap group in the staging buffer, FRAGMENT order: a tap is 2 halves x 32 columns x 16 floats; the sixteen floats of (half h, column n)
// are [kq'][u] with kq' = (kq + 2 ((n >> 3) & 1)) mod 4 (the swizzle keeps the sixteen lanes of a ds_read_b128 lane group on distinct
// 16-byte bank groups) and hold the contraction elements 16 h + 4 u + kq: forward B((j, c), n) = W[j][c][n] with c the contraction,
// input gradient B((j, o), c) = W[j][c][o] with o the contraction and c the column.  A lane's operand for a half-tap is one 16-byte read
// at a per-lane offset plus a wave-uniform one.
template <class P, class P4>
struct FxtConvW4 {
    P w;
    FXT_HD int prep(int n, int kq) const { return n * 16 + 4 * ((kq + 2 * ((n >> 3) & 1)) & 3); }
    FXT_HD float at(int st, int j, int k0) const { return w[st + (j * 2 + (k0 >> 4)) * 512 + ((k0 >> 2) & 3)]; }       // (scalar form: host build)
    FXT_HD auto at4(int st, int j, int h) const { return *(P4)(w + st + (j * 2 + h) * 512); }
};
// where element (row = jrel 32 + c, column e) of a kernel's tap group goes.  ROT 0: forward (contraction = row's channel c, column n = e);
// 1: input gradient (contraction = e, column = c)
template <int ROT>
FXT_HD int fxt_w4_off(int row, int e) {
    const int jrel = row >> 5, c = row & 31;
    const int k = ROT ? e : c, n = ROT ? c : e;              // contraction element, column
    return ((jrel * 2 + (k >> 4)) * 32 + n) * 16 + 4 * (((k & 3) + 2 * ((n >> 3) & 1)) & 3) + ((k >> 2) & 3);
}
FXT_HD bool fxt_conv32_ok(int Md, int F, int nw) { return F == 32 && ((Md + 15) >> 4) <= nw; }
// A tap group on its way from L2 to the staging buffer: PF 16-byte pieces per thread in registers.  Carried ACROSS products and phase
// barriers -- conv2's group is fetched while conv1 runs, conv3's first group behind conv2's MFMAs, conv3's input-gradient group while
// the max-pool backward runs, conv2's behind conv3's input gradient -- so that no product starts by waiting for L2, and (backward) no
// weight fetch is issued behind a phase's gradient-partial stores: vmcnt retires in order, a load issued after 78 KiB of partial stores
// waits for all of them (round 5: conv2's backward phase took 30 us for ~10 us of work in every form; this was why).
#define FXT_TAP_PF 2
// A piece = the sixteen bytes one lane of the product will read as ONE operand: piece p of a group is (tap p >> 8, half (p >> 7) & 1,
// column (p >> 2) & 31, lane group kq = p & 3) and holds the contraction elements 16 half + 4 u + kq, u = 0 .. 3 -- four dword loads
// (64-byte runs across the lanes) and ONE ds_write_b128 into the fragment order of FxtConvW4, conflict-free.  (Fetched as 16-byte
// row pieces and scattered by four ds_write_b32 the stores hit the banks 8-way: ~0.9 us per group between two barriers, round 5.)
template <int WAS>
struct FxtTapRegs {
#if FXT_DEVICE
    fxt_f4 pre[FXT_TAP_PF];
#endif
    int taps;                                              // taps held (0 = nothing)
    // ROT 0: forward (contraction = the kernel's input channel, column = output channel); 1: input gradient (the other way round)
    template <int ROT>
    FXT_HD void fetch(const FxtWg& wg, typename FxtMem<WAS>::CF wsrc, int g0, int g1) {      // taps [g0, g1) of a 32 x 32-per-tap kernel
        taps = g1 - g0;
#if FXT_DEVICE
        const int pieces = taps * 256;
#pragma unroll
        for (int q = 0; q < FXT_TAP_PF; ++q) {
            const int p = wg.tid + q * wg.nthr;
            if (p < pieces) {
                const int jr = p >> 8, half = (p >> 7) & 1, col = (p >> 2) & 31, kq = p & 3;
                typename FxtMem<WAS>::CF src = wsrc + (g0 + jr) * 1024 + (ROT ? col * 32 + half * 16 + kq : (half * 16 + kq) * 32 + col);
#pragma unroll
                for (int u = 0; u < 4; ++u) pre[q][u] = src[(ROT ? 4 : 128) * u];
            }
        }
#else
        (void)wg; (void)wsrc; (void)g0;
#endif
    }
    template <int WSAS, class WB>
    FXT_HD void store(const FxtWg& wg, WB wbuf) {
#if FXT_DEVICE
        const int pieces = taps * 256;
#pragma unroll
        for (int q = 0; q < FXT_TAP_PF; ++q) {
            const int p = wg.tid + q * wg.nthr;
            if (p < pieces) {
                const int blk = p >> 2, col = blk & 31, kq = p & 3;         // blk = (tap 2 + half) 32 + column
                *(typename FxtMem<WSAS>::F4)(wbuf + blk * 16 + 4 * ((kq + 2 * ((col >> 3) & 1)) & 3)) = pre[q];
            }
        }
#else
        (void)wg; (void)wbuf;
#endif
        taps = 0;
    }
};
// taps per group: what the host sized the buffer for, and what FXT_TAP_PF pieces per thread carry
FXT_HD int fxt_conv32_group(int stage_taps, int nthr) {
    const int gmax = FXT_TAP_PF * nthr / 256;
    int G = stage_taps < gmax ? stage_taps : gmax;
    return G < 1 ? 1 : G;
}
// ROT: 0 = forward rotation (16 c), 1 = input-gradient rotation (2 c).  `G` = fxt_conv32_group taps per group.
// tap: the register carrier.  pre_loaded: it holds taps [0, min(G, Ko)) already.  in_wbuf: those taps are in wbuf already (stored and
// published by an earlier barrier).  next_src / next_Ko: the NEXT product's kernel -- its first group is fetched into `tap` behind this
// product's last group of MFMAs and left there.
struct FxtNoDbg { FXT_HD void operator()(int) const {} };
template <int WSAS, int WAS, int ROT, class FA, class FB, class FC, class DBG = FxtNoDbg>
FXT_HD void fxt_conv32_staged(const FxtWg& wg, int Md, int Ko, const FA& fa, const FB& fb, const FC& fc,
                              typename FxtMem<WAS>::CF wsrc, typename FxtMem<WSAS>::F wbuf, int G, FxtTapRegs<WAS>& tap,
                              bool pre_loaded = false, bool in_wbuf = false, typename FxtMem<WAS>::CF next_src = nullptr, int next_Ko = 0,
                              const DBG& dbg = DBG()) {
    if (G < 1) G = 1;
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8;
    const int lane = wg.tid & 63, wave = fxt_wave(wg);
    const int i = lane & 15, kq = lane >> 4;
    const int tm = (Md + 15) >> 4;
    const bool have = wave < tm;                           // (wave-uniform; fxt_conv32_ok: every M tile has its wave)
    const int m = wave * 16 + i;
    const auto sa = fa.prep((have && m < Md) ? m : 0, kq);
    const auto sb0 = fb.prep(i, kq), sb1 = fb.prep(i + 16, kq);
    f4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const auto e0 = fc.pre(i), e1 = fc.pre(i + 16);        // what the epilogue needs per column (a bias from global memory): fetched now, not in a dependent chain at the end
    // half a tap's operands: four k-steps of A and of both B tiles.  Half-tap h + 1's are fetched BEFORE half-tap h's eight MFMAs (two
    // register sets of twelve, the loop unrolled by two): an LDS round trip hides behind ~256 cycles of matrix work instead of
    // preceding it.  (Whole taps in flight -- 2 x 24 registers -- spilled at the 128 registers sixteen waves leave a thread.)
    constexpr int UH = U / 2;
    struct Ops { f4_t a, b0, b1; };                        // (one 16-byte LDS read each: fxt_xi<2>, FxtConvW4)
    auto load = [&](Ops& o, int h, int g0) {               // half-tap h of the group: tap g0 + h / 2, k-steps 4 (h & 1) ...
        const int ko = g0 + (h >> 1);
        o.a = fa.at4(sa, ko, h & 1); o.b0 = fb.at4(sb0, ko - g0, h & 1); o.b1 = fb.at4(sb1, ko - g0, h & 1);
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[u], o.b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[u], o.b1[u], acc1, 0, 0, 0);
        }
    };
    if (!pre_loaded && !in_wbuf) tap.template fetch<ROT>(wg, wsrc, 0, G < Ko ? G : Ko);
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        if (!(in_wbuf && g0 == 0)) {
            fxt_sync_ws<WSAS>();                           // everybody is through with the previous group's taps (or the previous phase)
            tap.template store<WSAS>(wg, wbuf);
        }
        if (g1 < Ko) tap.template fetch<ROT>(wg, wsrc, g1, g1 + G < Ko ? g1 + G : Ko);       // in flight behind this group's MFMAs
        else if (next_src) tap.template fetch<ROT>(wg, next_src, 0, G < next_Ko ? G : next_Ko);
        fxt_sync_ws<WSAS>();
        dbg(2 * (g0 / G));                                 // (profiling aid: this wave's clock before / after a group's MFMAs)
        if (!have) continue;
        Ops x, y;
        const int H = 2 * (g1 - g0);                       // (even)
        load(x, 0, g0);
        for (int h = 0; h < H; h += 2) {
            load(y, h + 1, g0);
            FXT_SCHED_FENCE();                             // (the fetches stay IN FRONT of the MFMAs they hide behind: left alone, the scheduler sinks each one to just before its use)
            mma(x);
            load(x, h + 2 < H ? h + 2 : h, g0);            // (past the group's end: the same half-tap again, unused -- a straight-line body lets the waits be counted exactly)
            FXT_SCHED_FENCE();
            mma(y);
        }
        dbg(2 * (g0 / G) + 1);
    }
    if (have) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = wave * 16 + 4 * kq + r;
            if (mr < Md) { fc.put(mr, i, acc0[r], e0); fc.put(mr, i + 16, acc1[r], e1); }
        }
    }
#else
    (void)wg; (void)tap; (void)pre_loaded; (void)in_wbuf; (void)next_src; (void)next_Ko;     // (the host build stages every group itself: the same values)
    float* accs = new float[(size_t)Md * 32]();
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        for (int row = 0; row < (g1 - g0) * 32; ++row)
            for (int c = 0; c < 32; ++c) wbuf[fxt_w4_off<ROT>(row, c)] = wsrc[(g0 * 32 + row) * 32 + c];
        for (int m = 0; m < Md; ++m)
            for (int n = 0; n < 32; ++n) {
                float acc = accs[(size_t)m * 32 + n];
                for (int ko = g0; ko < g1; ++ko)
                    for (int ki = 0; ki < 32; ++ki)
                        acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko - g0, ki & ~3), acc);
                accs[(size_t)m * 32 + n] = acc;
            }
    }
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < 32; ++n) fc.put(m, n, accs[(size_t)m * 32 + n], fc.pre(n));
    delete[] accs;
#endif
}

// Conv weight gradient of a 32 -> 32 channel layer over ROTATED rows (see above): x, dz position-major arrays of R x L1 rows,
// fc.put(row (j 32 + c, or Kt 32 for the bias), column o, value).  Kt <= 20 taps.  `mid`: called by every wave ONCE, after its (first)
// job's loop and before any of its gradient stores (the caller commits a prefetched tap group there: a barrier inside).
// KT > 0: the tap count as a compile-time constant (canonical instantiations): a job's taps are then a constant per residue, its
// MFMAs unconditional, and the register window turns over by renaming inside blocks of five k-steps instead of by moves.
#define FXT_WG32_MAXT 5
struct FxtNoHook { FXT_HD void operator()() const {} };
#if FXT_DEVICE
typedef float fxt_acc4 __attribute__((ext_vector_type(4)));
// one job's loops: taps res, res + 4, ... (NT of them; NT < 0: a run-time count `ntr`), channel tile at c, out-channel tile at o
template <int NT, bool BIAS, class P>
FXT_HD void fxt_wg32_job(int R, int L1, int res, int pl, int c, int o, int kq, int ntr, bool biasr, P x, P dz, P zero,
                         fxt_acc4 (&acc)[FXT_WG32_MAXT], fxt_acc4& accb) {
    constexpr int MT = FXT_WG32_MAXT;
    const int S = (L1 + 3) >> 2, Sfull = L1 >> 2;          // k-steps of a row; those whose four positions all lie inside it
    for (int rho = 0; rho < R; ++rho) {
        const int base = rho * L1;
        // x block b: position 4 b + kq + res - pl of the row, channel c (zero outside the row).  Fetch and mask are separate steps:
        // the mask is applied where the value is USED, one k-step later, so nothing waits on LDS between a fetch and the MFMAs of
        // the k-step it is issued in.
        auto Xok = [&](int b) { const int pp = 4 * b + kq + res - pl; return pp >= 0 && pp < L1; };
        // (positions outside the row are read from `zero`, a row of zeros in LDS: no select behind the fetch)
        auto Xraw = [&](int b, bool ok) {
            const int pp = 4 * b + kq + res - pl;
            return *(ok ? x + fxt_xi<2>(base + pp, c, 32) : zero);
        };
        auto Braw = [&](int s, bool ok) {
            const int t = 4 * s + kq;
            return *(ok ? dz + fxt_xi<2>(base + t, o, 32) : zero);
        };
        float win[MT];                                     // win[(s + q) % MT] = x block s + q inside the blocks of MT k-steps below
#pragma unroll
        for (int q = 0; q < MT - 1; ++q) win[q] = Xraw(q, Xok(q));
        // two k-steps of operands in flight: (xr, br) for the step about to run, (xr2, br2) for the one after -- one step ahead covers
        // an LDS round trip only behind five MFMAs; conv2's jobs issue one or two per k-step (round 5: 345 cycles per k-step there)
        bool bok = kq < L1, bok2 = 4 + kq < L1;
        float xr = Xraw(MT - 1, Xok(MT - 1)), br = Braw(0, bok);
        float xr2 = Xraw(MT, Xok(MT)), br2 = Braw(1, bok2);
        int s = 0;
        if constexpr (NT >= 0) {
            for (; s + MT <= Sfull; s += MT) {             // MT whole k-steps: the window's slots are compile-time constants
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    win[(u + MT - 1) % MT] = xr;
                    const float b = br;
                    xr = xr2; br = br2; bok = bok2;
                    {   const int sn = s + u + 2;          // the fetches of the k-step after next (masked past the row's end)
                        bok2 = 4 * sn + kq < L1;
                        xr2 = Xraw(sn + MT - 1, Xok(sn + MT - 1)); br2 = Braw(sn, bok2); }
                    FXT_SCHED_FENCE();
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[(u + q) % MT], b, acc[q], 0, 0, 0);
                    if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, b, accb, 0, 0, 0);
                }
            }
        }
        const int nt = NT >= 0 ? NT : ntr;
        const bool bias = NT >= 0 ? BIAS : biasr;
        for (; s < S; ++s) {                               // the rest of the row (run-time tap counts: all of it): the window moves by copies
            win[MT - 1] = xr;
            const float b = br;
            const bool tok = bok;                          // 4 s + kq < L1: false only in a row's last, partial k-step -- both operands zero there, as fxt_gemm masks them
            xr = xr2; br = br2; bok = bok2;
            bok2 = 4 * (s + 2) + kq < L1;
            xr2 = Xraw(s + MT + 1, Xok(s + MT + 1)); br2 = Braw(s + 2, bok2);
#pragma unroll
            for (int q = 0; q < MT; ++q)
                if (q < nt) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(tok ? win[q] : 0.f, b, acc[q], 0, 0, 0);
            if (bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(tok ? 1.f : 0.f, b, accb, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MT - 1; ++q) win[q] = win[q + 1];
        }
    }
}
#endif
template <int KT = 0, class P, class FC, class HOOK = FxtNoHook>
FXT_HD void fxt_conv32_wgrad(const FxtWg& wg, int R, int L1, int Kt, int pl, P x, P dz, P zero, const FC& fc, const HOOK& mid = HOOK()) {
#if FXT_DEVICE
    constexpr int MT = FXT_WG32_MAXT;
    const int lane = wg.tid & 63, nw = wg.nthr >> 6;
    const int i = lane & 15, kq = lane >> 4;
    bool hooked = false;
    for (int job = fxt_wave(wg); job < 16; job += nw) {    // (sixteen waves: one job each)
        const int res = job >> 2, ct = (job >> 1) & 1, ot = job & 1;
        const int NT = Kt > res ? (Kt - res + 3) >> 2 : 0; // taps res, res + 4, ... of this job (wave-uniform)
        const bool bias = res == 3 && ct == 0;
        const int c = ct * 16 + i, o = ot * 16 + i;
        fxt_acc4 acc[MT], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < MT; ++q) acc[q] = fxt_acc4{0.f, 0.f, 0.f, 0.f};
        if constexpr (KT > 0) {
            constexpr int N0 = (KT + 3) >> 2, N1 = KT > 1 ? (KT + 2) >> 2 : 0, N2 = KT > 2 ? (KT + 1) >> 2 : 0, N3 = KT > 3 ? KT >> 2 : 0;
            if (res == 0) fxt_wg32_job<N0, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (res == 1) fxt_wg32_job<N1, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (res == 2) fxt_wg32_job<N2, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (bias) fxt_wg32_job<N3, true>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else fxt_wg32_job<N3, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
        } else {
            fxt_wg32_job<-1, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
        }
        if (!hooked) { mid(); hooked = true; }
#pragma unroll
        for (int q = 0; q < MT; ++q)
            if (q < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fc.put((res + 4 * q) * 32 + ct * 16 + 4 * kq + r, o, acc[q][r]);
            }
        if (bias && kq == 0) fc.put(Kt * 32, o, accb[0]);
    }
    if (!hooked) mid();                                    // (more than sixteen waves: the rest still meets the hook's barrier)
#else
    (void)wg; (void)zero;
    mid();
    for (int mrow = 0; mrow <= Kt * 32; ++mrow)
        for (int o = 0; o < 32; ++o) {
            const int j = mrow >> 5, c = mrow & 31;
            float acc = 0.f;
            for (int rho = 0; rho < R; ++rho)
                for (int t = 0; t < L1; ++t) {
                    const int pp = t + j - pl;
                    const float a = mrow == Kt * 32 ? 1.f : ((pp >= 0 && pp < L1) ? x[fxt_xi<2>(rho * L1 + pp, c, 32)] : 0.f);
                    acc = fmaf(a, dz[fxt_xi<2>(rho * L1 + t, o, 32)], acc);
                }
            fc.put(mrow, o, acc);
        }
#endif
}

