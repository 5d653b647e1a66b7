// NumPy's float32 summation order for one row of ensemble-member scores held in registers (np.mean / np.sum over
// axis 1 of a C-contiguous (N, M) array: flexs/ensemble.py:24,59): M < 8 sequential, else eight interleaved accumulators
// folded pairwise, then the tail.  Shared by the ensemble-mean kernels (misc_kernels.hip) and by the small-launch CNN
// kernel's fused mean (score_cnn_quad.hip), so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

// NumPy order for a compile-time row length (registers only)
template <int M>
__host__ __device__ __forceinline__ float np_sum_row(const float (&x)[M]) {
    if constexpr (M < 8) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) r += x[i];
        return r;
    } else {
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = x[k];
        constexpr int full = M - (M % 8);
#pragma unroll
        for (int i = 8; i < full; i += 8)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += x[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = full; i < M; ++i) res += x[i];
        return res;
    }
}

// round-to-nearest division on both sides (the device build runs with fast-math off; __fdiv_rn documents the intent)
__host__ __device__ __forceinline__ float fx_np_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

// mean of the first M entries of x (1 <= M <= 16), NumPy order, division as np.mean does it (sum / M in float32)
__host__ __device__ __forceinline__ float np_mean_row16(const float (&x)[16], int M) {
    switch (M) {
#define FX_NP_CASE(m) case m: { float y[m]; _Pragma("unroll") for (int i = 0; i < m; ++i) y[i] = x[i]; return fx_np_div(np_sum_row<m>(y), (float)m); }
        FX_NP_CASE(1) FX_NP_CASE(2) FX_NP_CASE(3) FX_NP_CASE(4) FX_NP_CASE(5) FX_NP_CASE(6) FX_NP_CASE(7) FX_NP_CASE(8)
        FX_NP_CASE(9) FX_NP_CASE(10) FX_NP_CASE(11) FX_NP_CASE(12) FX_NP_CASE(13) FX_NP_CASE(14) FX_NP_CASE(15) FX_NP_CASE(16)
#undef FX_NP_CASE
    }
    return 0.f;
}
