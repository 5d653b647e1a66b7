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

// One column step for one 64-bit block.  hin / return value in {-1, 0, +1}.
// top = bit index whose horizontal delta is reported (63, or (m-1) % 64 in the last block).
FX_HD int fx_myers_block(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, int hin, int top) {
    const uint64_t hneg = hin < 0 ? 1ull : 0ull;
    const uint64_t Xv = Eq | Mv;
    Eq |= hneg;
    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint64_t Ph = Mv | ~(Xh | Pv);
    uint64_t Mh = Pv & Xh;
    const int hout = (int)((Ph >> top) & 1ull) - (int)((Mh >> top) & 1ull);
    Ph <<= 1;
    Mh <<= 1;
    Mh |= hneg;
    Ph |= (uint64_t)((hin + 1) >> 1);
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

// Levenshtein(pattern, text) given the pattern's match masks.
//   peq(c, w): 64-bit mask, bit i set iff pattern[64*w + i] == c
//   text(i):   i-th text byte
// NULTERM: the text occupies a row of n bytes and ends at its first NUL byte (ragged rows are
// NUL-padded; no FLEXS alphabet contains NUL), so n is an upper bound of the text length.
template <int W, bool NULTERM = false, typename PeqFn, typename TextFn>
FX_HD int fx_myers_distance(int m, int n, PeqFn peq, TextFn text) {
    uint64_t Pv[W], Mv[W];
    const int nw = (m + 63) >> 6;
#pragma unroll
    for (int w = 0; w < W; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
    int score = m;
    const int top_last = (m - 1) & 63;
    for (int i = 0; i < n; ++i) {
        const int c = text(i);
        if (NULTERM && c == 0) break;
        int h = 1;                                   // D[0][j] - D[0][j-1] = +1 (global alignment)
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (w < nw) h = fx_myers_block(Pv[w], Mv[w], peq(c, w), h, (w == nw - 1) ? top_last : 63);
        }
        score += h;                                  // m == 0: no block runs, h stays +1 -> score = |text|
    }
    return score;
}
