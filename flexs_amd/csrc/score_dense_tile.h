// One 16-sequence tile of the MLP / GlobalEpistasis model walked by a GROUP OF 8 WAVES: wave w8 owns output tiles
// {w8, w8 + 8} of every layer, the layers' outputs meet in 2 x HT KiB of LDS (`hx`).  The explorer-size kernel
// (score_dense_small.hip) is this routine once per workgroup; the persistent kernel (score_dense_mfma.hip) runs it on the
// LAST tiles of a workgroup that do not divide among its four SIMDs (two groups of 8 of its 16 waves, two tiles at a
// time), reading the weights from its LDS image instead of global memory.  Every output element sees the arithmetic of
// the persistent kernel's one-wave walk (first layer: bias + the same rows -- pre-summed pair rows where that kernel uses
// them -- in position order; hidden layers: (input tile, k-step) order with the same tail skip; the same final dot), so
// the scores are the SAME BITS (tested).
//
// Every wave of the WORKGROUP must call it (the barriers are workgroup barriers); `live` = this wave's group has a tile.
// OVERLAY: `hx` lies over the first-layer rows (LDS is full), so one more barrier separates the last read of those rows
// from the first write of a layer output.
#pragma once
#include "fx_common.h"
#include "mfma_common.h"

template <int KIND, int HT, bool OVERLAY>
__device__ __forceinline__ void fx_dense_tile8(bool live, int w8, int lane, const uint8_t* row, int L, int A, int rlh, int pair,
                                               const float* w_first, const float* w1p, const float* w1pair, const f4* w_d2,
                                               const f4* w_d3, const float* db, const uint8_t* lut_s, f4* hx,
                                               volatile int* bad_flag, bool& bad, float& y_out) {
    constexpr int SW = 8;
    constexpr int OT = (HT + SW - 1) / SW;                // output tiles per wave (1 or 2)
    constexpr int PF = 8;                                 // first-layer rows in flight per output tile
    const int g = lane >> 4;

    if constexpr (KIND == FX_GE) {
        // ---- GE layer 1: s = relu(b1 + sum_l w1[l * A + code_l]), a scalar per sequence (every wave sums it itself: lane
        //      group g takes positions g, g + 4, ... in order, two cross-lane adds -- the persistent kernel's order);
        //      layer 2: relu(b2 + s * w2) for this wave's output tiles, directly in B-operand layout
        f4 v2[OT];
        if (live) {
            const unsigned amax = (unsigned)A - 1u;
            float sacc = 0.f;
            unsigned seen = 0;
            for (int l0 = g; l0 < L; l0 += 4 * PF) {
                float r[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int l = l0 + 4 * k;
                    if (l < L) {
                        const unsigned c = lut_s[row[l]];
                        seen |= c;
                        r[k] = w_first[l * A + (int)(c < amax ? c : amax)];
                    }
                }
#pragma unroll
                for (int k = 0; k < PF; ++k)
                    if (l0 + 4 * k < L) sacc += r[k];
            }
            bad |= seen >= 0x80u;
            if (bad_flag && bad) *bad_flag = 1;               // (read by the wave that answers, barriers later)
            sacc += __shfl_xor(sacc, 16);
            sacc += __shfl_xor(sacc, 32);
            sacc += db[0];
            const float sv = relu1(sacc);
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const int mo = w8 + SW * t;
                if (mo < HT) {
                    const f4 w2 = *reinterpret_cast<const f4*>(&db[4 + 16 * mo + 4 * g]);
                    const f4 b2 = *reinterpret_cast<const f4*>(&db[4 + 16 * HT + 16 * mo + 4 * g]);
                    f4 v;
                    v.x = relu1(fmaf(sv, w2.x, b2.x));
                    v.y = relu1(fmaf(sv, w2.y, b2.y));
                    v.z = relu1(fmaf(sv, w2.z, b2.z));
                    v.w = relu1(fmaf(sv, w2.w, b2.w));
                    v2[t] = v;
                }
            }
        }
        if constexpr (OVERLAY) __syncthreads();
        if (live) {
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const int mo = w8 + SW * t;
                if (mo < HT) hx[mo * 64 + lane] = v2[t];
            }
        }
        __syncthreads();
    } else {
        // ---- layer 1: relu(b1 + sum of the kernel rows selected by the codes), this wave's output tiles only
        f4 h[OT];
        if (live) {
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const int mo = w8 + SW * t;
                h[t] = mo < HT ? *reinterpret_cast<const f4*>(&db[16 * mo + 4 * g]) : splat4(0.f);
            }
            unsigned seen = 0;
            if (pair == 2) {
                // the tile's first layer was taken by k_mlp_l1_pos (score_dense_l1.h): `w1pair` is its [mo][64 lanes] block (relu'd; the relu below is idempotent)
#pragma unroll
                for (int t = 0; t < OT; ++t) {
                    const int mo = w8 + SW * t;
                    if (mo < HT) h[t] = reinterpret_cast<const f4*>(w1pair)[mo * 64 + lane];
                }
            } else if (pair) {
                const float* wp = w1pair + 4 * g;
                constexpr int RS = 16 * HT + FX_PAIR_PAD;
                const int np2 = L >> 1, nterm = np2 + (L & 1);    // pair rows, then the odd last position's own row
                for (int t0 = 0; t0 < nterm; t0 += PF) {
                    f4 r[PF][OT];
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int pi = t0 + k;
                        if (pi < nterm) {
                            int rowi;
                            if (pi < np2) {
                                const unsigned c0 = lut_s[row[2 * pi]], c1 = lut_s[row[2 * pi + 1]];
                                seen |= c0 | c1;
                                rowi = pi * 16 + (int)(((c0 & 3u) << 2) | (c1 & 3u));
                            } else {
                                const unsigned c0 = lut_s[row[L - 1]];
                                seen |= c0;
                                rowi = np2 * 16 + (int)(c0 & 3u);
                            }
#pragma unroll
                            for (int t = 0; t < OT; ++t) {
                                const int mo = w8 + SW * t;
                                if (mo < HT) r[k][t] = *reinterpret_cast<const f4*>(wp + (int64_t)rowi * RS + 16 * mo);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < PF; ++k)
                        if (t0 + k < nterm)
#pragma unroll
                            for (int t = 0; t < OT; ++t) h[t] += r[k][t];
                }
            } else {
                const float* w1 = w1p + 4 * g;
                const unsigned amax = (unsigned)A - 1u;
                for (int l0 = 0; l0 < L; l0 += PF) {
                    f4 r[PF][OT];
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int l = l0 + k;
                        if (l < L) {
                            const unsigned c = lut_s[row[l]];
                            seen |= c;
                            const unsigned ci = c < amax ? c : amax;   // keeps the read inside the table for a bad character
#pragma unroll
                            for (int t = 0; t < OT; ++t) {
                                const int mo = w8 + SW * t;
                                if (mo < HT) r[k][t] = *reinterpret_cast<const f4*>(w1 + ((int64_t)l * A + ci) * (16 * HT) + 16 * mo);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < PF; ++k)
                        if (l0 + k < L)
#pragma unroll
                            for (int t = 0; t < OT; ++t) h[t] += r[k][t];
                }
            }
            bad |= seen >= 0x80u;
            if (bad_flag && bad) *bad_flag = 1;
        }
        if constexpr (OVERLAY) __syncthreads();
        if (live) {
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const int mo = w8 + SW * t;
                if (mo < HT) hx[mo * 64 + lane] = relu4(h[t]);
            }
        }
        __syncthreads();
    }

    // ---- hidden HxH layers: this wave's output tiles from all HT input tiles
    auto hidden = [&](const f4* wblk, const float* bias, const f4* src, f4* dst) {
        f4 in[HT];
#pragma unroll
        for (int mi = 0; mi < HT; ++mi) in[mi] = src[mi * 64 + lane];
#pragma unroll
        for (int t = 0; t < OT; ++t) {
            const int mo = w8 + SW * t;
            if (mo < HT) {
                f4 a[HT];
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) a[mi] = wblk[(mi * HT + mo) * 64 + lane];
                f4 acc = *reinterpret_cast<const f4*>(&bias[16 * mo + 4 * g]);
#pragma unroll
                for (int mi = 0; mi < HT; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mi == HT - 1 && r >= rlh) break;
                        acc = mfma16(a[mi][r], in[mi][r], acc);
                    }
                dst[mo * 64 + lane] = relu4(acc);
            }
        }
    };
    const f4* last = hx;                                 // where the last hidden layer's output ends up
    if constexpr (KIND == FX_GE) {
        if (live) hidden(w_d3, db + 4 + 32 * HT, hx, hx + HT * 64);
        last = hx + HT * 64;
    } else {
        if (live) hidden(w_d2, db + 16 * HT, hx, hx + HT * 64);
        __syncthreads();
        if (live) hidden(w_d3, db + 32 * HT, hx + HT * 64, hx);
    }
    __syncthreads();

    // ---- Dense(1): the group's first wave
    if (live && w8 == 0) {
        f4 h3[HT][1];
#pragma unroll
        for (int mi = 0; mi < HT; ++mi) h3[mi][0] = last[mi * 64 + lane];
        float y[1];
        if constexpr (KIND == FX_GE) final_dot<HT, 1>(db + 4 + 48 * HT, db[4 + 64 * HT], h3, y, g);
        else final_dot<HT, 1>(db + 48 * HT, db[64 * HT], h3, y, g);
        y_out = y[0];
    }
}
