// Bit-parallel unit-cost Levenshtein distance (Myers 1999 in Hyyro's 2003
// edit-distance formulation, multi-word blocks with horizontal carries).
// Replaces `editdistance.eval` (third-party C++, call site
// noisy_abstract_model.py:51) for the NoisyAbstractModel neighbour search.
// Shared by the device kernel (mindist.hip) and a host debug entry point so the
// exact same code is property-tested on CPU against the DP oracle.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FX_HD __host__ __device__ __forceinline__
#else
#define FX_HD inline
#endif

// One column step for one block of `8 * sizeof(Word)` pattern rows.  hin / return value in {-1, 0, +1}.
// top = bit index whose horizontal delta is reported (the block's last bit, or (m-1) % bits in the last block).
// Word = uint32_t halves the integer work for patterns of <= 32 symbols (the DNA / RNA landscapes).
template <typename Word>
FX_HD int fx_myers_block(Word& Pv, Word& Mv, Word Eq, int hin, int top) {
    const Word hneg = hin < 0 ? Word(1) : Word(0);
    const Word Xv = Eq | Mv;
    Eq |= hneg;
    const Word Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    Word Ph = Mv | ~(Xh | Pv);
    Word Mh = Pv & Xh;
    const int hout = (int)((Ph >> top) & Word(1)) - (int)((Mh >> top) & Word(1));
    Ph <<= 1;
    Mh <<= 1;
    Mh |= hneg;
    Ph |= (Word)((hin + 1) >> 1);
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

// Levenshtein(pattern, text) given the pattern's match masks.
//   peq(c, w): Word mask, bit i set iff pattern[bits*w + i] == c
//   text(i):   i-th text byte
// NULTERM: the text occupies a row of n bytes and ends at its first NUL byte (ragged rows are
// NUL-padded; no FLEXS alphabet contains NUL), so n is an upper bound of the text length.
template <int W, bool NULTERM = false, typename Word = uint64_t, typename PeqFn, typename TextFn>
FX_HD int fx_myers_distance(int m, int n, PeqFn peq, TextFn text) {
    constexpr int BITS = 8 * (int)sizeof(Word);
    Word Pv[W], Mv[W];
    const int nw = (m + BITS - 1) / BITS;
#pragma unroll
    for (int w = 0; w < W; ++w) { Pv[w] = ~Word(0); Mv[w] = Word(0); }
    int score = m;
    const int top_last = (m - 1) & (BITS - 1);
    for (int i = 0; i < n; ++i) {
        const int c = text(i);
        if (NULTERM && c == 0) break;
        int h = 1;                                   // D[0][j] - D[0][j-1] = +1 (global alignment)
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (w < nw) h = fx_myers_block<Word>(Pv[w], Mv[w], (Word)peq(c, w), h, (w == nw - 1) ? top_last : BITS - 1);
        }
        score += h;                                  // m == 0: no block runs, h stays +1 -> score = |text|
    }
    return score;
}

// ---- patterns longer than one register block set: the recurrence in STRIPS -------------------------------------------
// Strip s = pattern rows [64 LW s, 64 LW (s + 1)) run over all text columns.  Its top boundary is the bottom boundary
// of strip s - 1 -- one horizontal delta in {-1, 0, +1} per column, read from `hin` and overwritten in place through
// `hout` (element i at [i * stride]) -- its left boundary the usual +1 vertical deltas (D[i][0] = i).  The last strip
// returns the sum of its bottom deltas: distance = m + that sum.  `rows` = pattern rows in this strip (0 for an empty
// pattern: every column then contributes +1, i.e. the distance is the text length).
template <int LW, typename PeqFn, typename TextFn>
FX_HD int fx_myers_strip(int rows, int n, PeqFn peq, TextFn text, const signed char* hin, signed char* hout, size_t stride,
                         bool first, bool last) {
    uint64_t Pv[LW], Mv[LW];
    const int nw = (rows + 63) / 64;
#pragma unroll
    for (int w = 0; w < LW; ++w) { Pv[w] = ~uint64_t(0); Mv[w] = uint64_t(0); }
    const int top_last = (rows - 1) & 63;
    int sum = 0;
    for (int i = 0; i < n; ++i) {
        const int c = text(i);
        if (c == 0) break;                                  // NUL-padded (ragged) row
        int h = first ? 1 : (int)hin[(size_t)i * stride];
#pragma unroll
        for (int w = 0; w < LW; ++w)
            if (w < nw) h = fx_myers_block<uint64_t>(Pv[w], Mv[w], (uint64_t)peq(c, w), h, (w == nw - 1) ? top_last : 63);
        if (last) sum += h; else hout[(size_t)i * stride] = (signed char)h;
    }
    return sum;
}

// min(distance, K + 1) of the pattern qs[0 .. m) and the text row t[0 .. L) (NUL-padded when shorter): Ukkonen's band, 2 K + 1 cells
// per text column, left as soon as a whole band column exceeds K.  hamming: position-wise mismatches over the whole row instead.
template <int K>
FX_HD int fx_bounded_distance(bool hamming, int m, int L, const uint8_t* qs, const uint8_t* t) {
    constexpr int B = 2 * K + 1, INF = K + 1;
    if (hamming) {
        int d = 0;
        for (int i = 0; i < L && d <= K; ++i) d += (t[i] != qs[i]);
        return d > K ? INF : d;
    }
    int prev[B];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < B; ++r) { const int j = r - K; prev[r] = (j >= 0 && j <= m) ? (j < INF ? j : INF) : INF; }      // row 0: D[0][j] = j
    int n = L;
    for (int i = 1; i <= L; ++i) {
        const int bc = t[i - 1];
        if (bc == 0) { n = i - 1; break; }                    // NUL-padded (ragged) row
        int cur[B];
        int left = INF, best = INF;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int r = 0; r < B; ++r) {
            const int j = i + r - K;
            int v = INF;
            if (j == 0) v = i < INF ? i : INF;               // D[i][0] = i
            else if (j > 0 && j <= m) {
                const int diag = prev[r] + (qs[j - 1] != bc);
                const int up = r + 1 < B ? prev[r + 1] + 1 : INF;
                v = diag < up ? diag : up;
                v = left + 1 < v ? left + 1 : v;
                v = v < INF ? v : INF;
            }
            cur[r] = v; left = v;
            best = v < best ? v : best;
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int r = 0; r < B; ++r) prev[r] = cur[r];
        if (best > K) return INF;
    }
    const int rr = m - n + K;
    int d = INF;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < B; ++r) if (r == rr) d = prev[r];
    return d;
}
