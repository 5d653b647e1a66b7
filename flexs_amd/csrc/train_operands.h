// Part of the training step (train_core.h includes it; not a stand-alone header): the operand index functors of fxt_gemm (plain, rotated and fragment rows) and the compile-time dimension packs.
#pragma once

// ---- operand functors ------------------------------------------------------------------------------------------
// A k-step covers contraction indices ki = k0 + kq, kq = lane >> 4 in 0..3, k0 wave-uniform.  Every functor splits its
// address into a per-lane part (`prep`, once per tile: row / column decomposition, the kq term) and a wave-uniform part
// built from (ko, k0) in `at` -- scalar arithmetic on the GPU -- so that an operand fetch costs one or two vector
// instructions.  (The first build recomputed the whole index per element: ~12 VALU instructions per operand, and with
// four waves per SIMD the address arithmetic, not the memory, set the step time.)
template <class P>
struct FxtRowMajorA {          // A(m, 0, ki) = p[m * ld + ki]
    P p; int ld;
    FXT_HD int prep(int m, int kq) const { return m * ld + kq; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0]; }
};
template <class P>
struct FxtRowMajorB {          // B(0, ki, n) = p[ki * ld + n]
    P p; int ld;
    FXT_HD int prep(int n, int kq) const { return kq * ld + n; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0 * ld]; }
};
template <class P>
struct FxtTransB {             // B(0, ki, n) = p[n * ld + ki]      (W^T for the input gradients)
    P p; int ld;
    FXT_HD int prep(int n, int kq) const { return n * ld + kq; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0]; }
};
// conv forward: rows m = (r, t), contraction (tap j, channel c): A = x[r][t + j - pl][c] inside the sequence, else 0
template <class P>
struct FxtConvA {               // (C = x's row stride)
    P x; int Lx, C, pl; FxtDiv dL;
    struct St { int base, tp; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{(m - pl) * C + kq, t - pl}; }
    FXT_HD float at(St s, int j, int k0) const {
        const int p = s.tp + j;
        const bool ok = p >= 0 && p < Lx;
        const float v = x[ok ? s.base + j * C + k0 : 0];             // (clamped index + select: no branch around the load)
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvW {              // B((j, c), n) = w[(j * C + c) * F + n]      (F = the kernel's row stride)
    P w; int C, F;
    FXT_HD int prep(int n, int kq) const { return kq * F + n; }
    FXT_HD float at(int st, int j, int k0) const { return w[st + (j * C + k0) * F]; }
};
// conv input gradient: rows m = (r, s), contraction (tap j, out channel o): A = dz[r][s - j + pl][o], B = w[j][n][o]
template <class P>
struct FxtConvGradA {           // (F = dz's row stride)
    P dz; int Lx, F, pl; FxtDiv dL;
    struct St { int base, sp; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{(m + pl) * F + kq, s + pl}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        const bool ok = p >= 0 && p < Lx;
        const float v = dz[ok ? st.base - j * F + k0 : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvGradW {          // B((j, o), n = c) = w[(j * C + c) * F + o]  (F = the kernel's row stride)
    P w; int C, F;
    FXT_HD int prep(int n, int kq) const { return n * F + kq; }
    FXT_HD float at(int st, int j, int k0) const { return w[st + j * C * F + k0]; }
};
// conv weight gradient: rows m = (tap j, channel c) plus ONE extra row for the bias; contraction (row r, position t)
template <class P>
struct FxtConvWGradA {
    P x; int Lx, C, ld, pl, rows; FxtDiv dC; // rows = taps * C (row `rows` is the bias row: all ones); ld = x's row stride
    struct St { int off, tp; };              // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0};
        const int j = fxt_quot(m, dC), c = m - j * C;
        return St{(j - pl + kq) * ld + c + (1 << 30), j - pl + kq};     // (+2^30: keeps `off` non-negative for taps left of the sequence)
    }
    FXT_HD float at(St s, int r, int k0) const {
        const int p = s.tp + k0;
        const bool ok = s.off >= 0 && p >= 0 && p < Lx;
        const float v = x[ok ? s.off - (1 << 30) + (r * Lx + k0) * ld : 0];
        return s.off < 0 ? 1.f : (ok ? v : 0.f);
    }
};
// conv1 / first dense layer: x is the one-hot of the codes.  Rows m = (j, c) = m / A, m % A plus the bias row.
// conv = 1: contraction (ko = row r, ki = position t), element [code[r][t + j] == c];
// conv = 0: contraction (ko = 0, ki = row r),          element [code[r][j] == c]   (j = the position of input unit m)
template <class P>
struct FxtOneHotWGradA {
    P codes; int L, A, rows, conv; FxtDiv dA;
    struct St { int off, c; };               // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0};
        const int j = fxt_quot(m, dA), c = m - j * A;
        return St{conv ? j + kq : kq * L + j, c};
    }
    FXT_HD float at(St s, int ko, int k0) const {
        const int code = codes[s.off < 0 ? 0 : s.off + (conv ? ko * L + k0 : k0 * L)];
        return (s.off < 0 || code == s.c) ? 1.f : 0.f;
    }
};
template <class P>
struct FxtPosMajorB {          // B((r, t), n) = p[(r * Lx + t) * F + n]     (F = p's row stride)
    P p; int Lx, F;
    FXT_HD int prep(int n, int kq) const { return kq * F + n; }
    FXT_HD float at(int st, int r, int k0) const { return p[st + (r * Lx + k0) * F]; }
};
// dense weight gradient: rows m = input unit k plus the bias row; contraction over the slice's rows r
template <class P>
struct FxtDenseWGradA {
    P in; int Kd, ld;          // ld = in's row stride
    FXT_HD int prep(int m, int kq) const { return m >= Kd ? -1 : kq * ld + m; }
    FXT_HD float at(int st, int, int k0) const { const float v = in[st < 0 ? 0 : st + k0 * ld]; return st < 0 ? 1.f : v; }
};

// ---- the same operands over ROTATED rows (fxt_xi<true>; the row stride equals the channel count, a power of two) ----
template <class P>
struct FxtConvASwz {            // element (row m - pl + j, channel k0 + kq)
    P x; int Lx, C, pl; FxtDiv dL;
    struct St { int base, tp, rot; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{(m - pl) * C, t - pl, kq + 2 * (m - pl)}; }
    FXT_HD float at(St s, int j, int k0) const {
        const int p = s.tp + j;
        const bool ok = p >= 0 && p < Lx;
        const float v = x[ok ? s.base + j * C + ((s.rot + 2 * j + k0) & (C - 1)) : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvGradASwz {        // element (row m + pl - j, channel k0 + kq)
    P dz; int Lx, F, pl; FxtDiv dL;
    struct St { int base, sp, rot; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{(m + pl) * F, s + pl, kq + 2 * (m + pl)}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        const bool ok = p >= 0 && p < Lx;
        const float v = dz[ok ? st.base - j * F + ((st.rot - 2 * j + k0) & (F - 1)) : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvWGradASwz {       // element (row r Lx + k0 + j - pl + kq, channel c)
    P x; int Lx, C, ld, pl, rows; FxtDiv dC;
    struct St { int off, tp, rot; };         // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0, 0};
        const int j = fxt_quot(m, dC), c = m - j * C;
        return St{(j - pl + kq) * ld + (1 << 30), j - pl + kq, c + 2 * (j - pl + kq)};
    }
    FXT_HD float at(St s, int r, int k0) const {
        const int p = s.tp + k0;
        const bool ok = s.off >= 0 && p >= 0 && p < Lx;
        const int ru = r * Lx + k0;
        const float v = x[ok ? s.off - (1 << 30) + ru * ld + ((s.rot + 2 * ru) & (ld - 1)) : 0];
        return s.off < 0 ? 1.f : (ok ? v : 0.f);
    }
};
template <class P>
struct FxtPosMajorBSwz {        // element (row r Lx + k0 + kq, channel n)
    P p; int Lx, F;
    struct St { int off, rot; };
    FXT_HD St prep(int n, int kq) const { return St{kq * F, n + 2 * kq}; }
    FXT_HD float at(St st, int r, int k0) const { const int ru = r * Lx + k0; return p[st.off + ru * F + ((st.rot + 2 * ru) & (F - 1))]; }
};

// MODE 3: the conv A operands over FRAGMENT rows (fxt_xi<2>), positions outside the row sent to a row of zeros in LDS (`zoff`: its index relative to the
// array) instead of a select on the fetched value -- the fetch then has no consumer but its MFMA, so it can be issued a half-tap ahead
// (a select right behind the fetch made the compiler wait for LDS there), and four instructions per half-tap go away.  0.0 either way.
template <class P, class P4>
struct FxtConvAZ {              // conv forward over fragment rows: element (row m - pl + j, channel k0 + kq)
    P x; int Lx, pl, zoff; FxtDiv dL;
    struct St { int row, tp, kq; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{m - pl, t - pl, kq}; }
    FXT_HD float at(St s, int j, int k0) const {             // (scalar form: host build)
        const int p = s.tp + j;
        return (p >= 0 && p < Lx) ? x[fxt_xi<2>(s.row + j, k0 + s.kq, 32)] : 0.f;
    }
    FXT_HD auto at4(St s, int j, int h) const {              // channels 16 h + 4 u + kq, u = 0 .. 3: one 16-byte read
        const int p = s.tp + j, row = s.row + j;
        const bool ok = p >= 0 && p < Lx;
        return *(P4)(x + (ok ? row * 32 + 4 * ((4 * h + s.kq + (row & 6)) & 7) : zoff));
    }
};
template <class P, class P4>
struct FxtConvGradAZ {          // conv input gradient over fragment rows: element (row m + pl - j, channel k0 + kq)
    P dz; int Lx, pl, zoff; FxtDiv dL;
    struct St { int row, sp, kq; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{m + pl, s + pl, kq}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        return (p >= 0 && p < Lx) ? dz[fxt_xi<2>(st.row - j, k0 + st.kq, 32)] : 0.f;
    }
    FXT_HD auto at4(St st, int j, int h) const {
        const int p = st.sp - j, row = st.row - j;
        const bool ok = p >= 0 && p < Lx;
        return *(P4)(dz + (ok ? row * 32 + 4 * ((4 * h + st.kq + (row & 6)) & 7) : zoff));
    }
};
// B((r, t), n) over any layout (conv1's weight gradient reads the fragment rows through the shape-agnostic product)
template <class P, int LAY>
struct FxtPosMajorBL {
    P p; int Lx, F;
    struct St { int n, kq; };
    FXT_HD St prep(int n, int kq) const { return St{n, kq}; }
    FXT_HD float at(St st, int r, int k0) const { return p[fxt_xi<LAY>(r * Lx + k0 + st.kq, st.n, F)]; }
};

// Compile-time shape of a CANONICAL network (round 4).  The step is written for any shape the constructors accept: every
// contraction chooses among three k-step walks at run time, masks its overhangs, and builds its addresses from run-time
// dimensions -- 100 KiB of code per placement, executed once per launch, i.e. streamed through the 64 KiB instruction cache
// every step, and ~30 non-MFMA instructions per MFMA (profiles/r3_train_pmc.md).  For the shapes the explorers' surrogates are
// actually built with (CNN(32, 100, kernel 5) on 4 letters, MLP(100), GlobalEpistasis(100)) the same
// source is instantiated with the dimensions as constants: dead walks and masks fold away, offsets become immediates.  Same
// arithmetic in the same order: the SAME BITS as the generic instantiation (GPU test).
struct FxtDimsAny { static constexpr bool fixed = false; static constexpr int kind = 0, A = 0, F = 0, H = 0, K = 0, R = 0, L = 0; };
template <int KIND, int A_, int F_, int H_, int K_, int R_, int L_ = 0>     // L_ = 0 / R_ = 0: the sequence length / the rows per slice stay run-time values
struct FxtDims { static constexpr bool fixed = true; static constexpr int kind = KIND, A = A_, F = F_, H = H_, K = K_, R = R_, L = L_; };

