// One 16-sequence tile of the canonical 4-letter CNN walked by a QUAD of waves (one per SIMD), layer by layer through two exchange buffers
// in LDS -- phases A ... F of score_cnn_quad.hip's round (see there), as a routine the persistent kernel (score_cnn_kernel.h, QT) runs on
// the LAST (tiles mod 4) tiles of a workgroup instead of giving one SIMD a tile more than the others (round 6).
//     wave q takes the conv positions q, q + 4, ...; dense output tiles {q, q + 4}; wave 0 the final dot
// Every output element sees exactly the MFMA / add sequence of the one-wave kernel (k-order (tap, input tile, k-step), padding taps
// skipped, hidden tail k-steps skipped), the exchanges are copies: the scores are the SAME BITS (tests/test_gpu_forms.py).
// Every wave of the WORKGROUP must call it (the five barriers are workgroup barriers); `live` = this wave's quad has a tile.
// X, Y: this quad's two buffers of XT KiB (XT >= FT * positions and >= HT + 1).
#pragma once
#include "fx_common.h"
#include "mfma_common.h"

template <int FT, int HT, int K, int L1C, int XT>
__device__ __forceinline__ void fx_cnn_quad_round(bool live, int q, int lane, const uint8_t* row, const uint8_t* lut_s,
                                                  const float* w1p, const float* cb, const f4* w_c2, const f4* w_c3, const f4* w_d1,
                                                  const f4* w_d2, const float* db, int rlh, f4* X, f4* Y, bool& bad, float& y_out) {
    constexpr int A = 4, K3 = 3, PL2 = (K - 1) / 2, PL3 = 1, L1 = L1C;
    static_assert(XT >= HT + 1 && XT >= FT * L1C, "the exchange buffers hold a layer's output tiles");
    const int g = lane >> 4;
    asm volatile("" ::: "memory");                           // keep the LDS weight reads inside the round
    // ---- A: conv1 (valid) at this wave's positions: bias + the K kernel rows selected by the codes, relu
    if (live) {
        for (int pos = q; pos < L1; pos += 4) {
            f4 o1[FT][1];
            int c[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                c[j] = lut_s[row[pos + j]];
                if (c[j] == 0xFF) { bad = true; c[j] = 0; }
            }
            init_bias<FT, 1>(cb, o1, g);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float* rowp = w1p + (j * A + c[j]) * FX_C1_ROW(FT) + 4 * g;
#pragma unroll
                for (int t = 0; t < FT; ++t) o1[t][0] += *reinterpret_cast<const f4*>(rowp + 16 * t);
            }
            relu_tiles<FT, 1>(o1);
#pragma unroll
            for (int t = 0; t < FT; ++t) X[(pos * FT + t) * 64 + lane] = o1[t][0];
        }
    }
    __syncthreads();
    // ---- B: conv2 (same) at this wave's positions; tap j reads out1[pos + j - PL2], padding taps contribute nothing
    if (live) {
        for (int pos = q; pos < L1; pos += 4) {
            asm volatile("" ::: "memory");
            f4 o2[FT][1];
            init_bias<FT, 1>(cb + 16 * FT, o2, g);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int pp = pos + j - PL2;
                if (pp >= 0 && pp < L1) {
                    f4 in[FT][1];
#pragma unroll
                    for (int t = 0; t < FT; ++t) in[t][0] = X[(pp * FT + t) * 64 + lane];
                    mma_layer<FT, FT, 1>(w_c2 + j * FT * FT * 64, in, o2, lane);
                }
            }
            relu_tiles<FT, 1>(o2);
#pragma unroll
            for (int t = 0; t < FT; ++t) Y[(pos * FT + t) * 64 + lane] = o2[t][0];
        }
    }
    __syncthreads();
    // ---- C: conv3 (same, 3 taps) at this wave's positions from out2[pos + j - PL3]; relu through the pooled max with 0
    if (live) {
        for (int pos = q; pos < L1; pos += 4) {
            asm volatile("" ::: "memory");
            f4 o3[FT][1];
            init_bias<FT, 1>(cb + 32 * FT, o3, g);
#pragma unroll
            for (int j = 0; j < K3; ++j) {
                const int pp = pos + j - PL3;
                if (pp >= 0 && pp < L1) {
                    f4 in[FT][1];
#pragma unroll
                    for (int t = 0; t < FT; ++t) in[t][0] = Y[(pp * FT + t) * 64 + lane];
                    mma_layer<FT, FT, 1>(w_c3 + j * FT * FT * 64, in, o3, lane);
                }
            }
#pragma unroll
            for (int t = 0; t < FT; ++t) {
                o3[t][0] = pool_max4(splat4(0.f), o3[t][0]);
                X[(pos * FT + t) * 64 + lane] = o3[t][0];
            }
        }
    }
    __syncthreads();
    // ---- D: GlobalMaxPooling1D over all positions; dense 1 for this wave's output tiles {q, q + 4}
    if (live) {
        f4 gmax[FT][1];
#pragma unroll
        for (int t = 0; t < FT; ++t) {
            gmax[t][0] = splat4(0.f);
            for (int pp = 0; pp < L1; ++pp) gmax[t][0] = pool_max4(gmax[t][0], X[(pp * FT + t) * 64 + lane]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int mo = q + 4 * k;
            if (mo < HT) {
                f4 acc = *reinterpret_cast<const f4*>(&db[16 * mo + 4 * g]);
#pragma unroll
                for (int mi = 0; mi < FT; ++mi) {
                    const f4 a = w_d1[(mi * HT + mo) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma16(a[r], gmax[mi][0][r], acc);
                }
                Y[mo * 64 + lane] = relu4(acc);
            }
        }
    }
    __syncthreads();
    // ---- E: dense 2 for the output tiles {q, q + 4} from all HT tiles of dense 1
    if (live) {
        f4 h1[HT];
#pragma unroll
        for (int mi = 0; mi < HT; ++mi) h1[mi] = Y[mi * 64 + lane];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int mo = q + 4 * k;
            if (mo < HT) {
                f4 acc = *reinterpret_cast<const f4*>(&db[16 * HT + 16 * mo + 4 * g]);
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) {
                    const f4 a = w_d2[(mi * HT + mo) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mi == HT - 1 && r >= rlh) break;
                        acc = mfma16(a[r], h1[mi][r], acc);
                    }
                }
                X[mo * 64 + lane] = relu4(acc);
            }
        }
    }
    __syncthreads();
    // ---- F: Dense(1) on wave 0 of the quad
    if (live && q == 0) {
        f4 h2[HT][1];
#pragma unroll
        for (int mi = 0; mi < HT; ++mi) h2[mi][0] = X[mi * 64 + lane];
        float y[1];
        final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2, y, g);
        y_out = y[0];
    }
}
