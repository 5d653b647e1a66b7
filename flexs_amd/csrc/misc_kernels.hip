// HBM-bound helper kernels: stand-alone encode (K-enc), ensemble reduce (K3),
// NoisyAbstractModel blend (K5), argmax decode (K6).
#include "fx_common.h"
#include "mfma_common.h"

namespace {

// ---- string_to_one_hot over a batch (sequence_utils.py:32-47, keras_model.py:70-75).
// One thread per (sequence, position): reads 1 byte, writes A floats (one
// 16-byte store when A == 4).  Algorithmic bytes per sequence: L + 4*L*A.
template <int AT>
__global__ void k_encode_onehot(const uint8_t* __restrict__ ascii, const uint8_t* __restrict__ lut,
                                int64_t rows, int A, float* __restrict__ out, unsigned* err) {
    __shared__ uint8_t lut_s[256];
    if (threadIdx.x < 64)
        reinterpret_cast<uint32_t*>(lut_s)[threadIdx.x] = reinterpret_cast<const uint32_t*>(lut)[threadIdx.x];
    __syncthreads();
    bool bad = false;
    if (AT == 4) {
        // A == 4: one 16-byte store per position
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
            int c = lut_s[ascii[i]];
            if (c == 0xFF) { bad = true; c = -1; }
            reinterpret_cast<float4*>(out)[i] = make_float4(c == 0, c == 1, c == 2, c == 3);
        }
    } else if (AT == 1 || AT == 20) {
        // A % 4 == 0: one thread per 16-byte quad of the output, consecutive lanes -> consecutive quads
        // (AT == 20: protein alphabet, quads per position known at compile time -> no runtime division)
        const int q = AT == 20 ? 5 : (A >> 2);
        const int64_t quads = rows * q;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t pos = i / q;
            const int a4 = (int)(i - pos * q);
            int c = lut_s[ascii[pos]];
            if (c == 0xFF) { bad = true; c = -1; }
            const int k = c - 4 * a4;
            reinterpret_cast<float4*>(out)[i] = make_float4(k == 0, k == 1, k == 2, k == 3);
        }
    } else {
        const int64_t total = rows * A;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t pos = i / A;
            const int a = (int)(i - pos * A);
            const int c = lut_s[ascii[pos]];
            if (c == 0xFF) bad = true;
            out[i] = (a == c) ? 1.f : 0.f;
        }
    }
    if (bad) fx_raise(err, FX_ERR_BADCHAR);
}

// ---- NumPy pairwise summation order of one contiguous row (ensemble.py:24:
// np.mean(x, axis=1) on the stacked float32 matrix).  Restated from NumPy's
// pairwise_sum; verified bit-exact against numpy 2.2 (tests/test_oracle.py).
template <typename T, typename Load>
__device__ T np_pairwise(Load ld, int64_t base, int n) {
    if (n < 8) {
        T r = T(0);
        for (int i = 0; i < n; ++i) r += ld(base + i);
        return r;
    }
    if (n <= 128) {
        T r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = ld(base + k);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += ld(base + i + k);
        }
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += ld(base + i);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise<T>(ld, base, n2) + np_pairwise<T>(ld, base + n2, n - n2);
}

#include "np_sum.h"

// M <= 16: each thread reduces 4 consecutive rows = 4*M contiguous floats = M 16-byte loads,
// and writes one 16-byte result; HBM-bound (4*M + 4 bytes per sequence).
template <int M>
__global__ void k_ensemble_mean_small(const float* __restrict__ s, int64_t N, float* __restrict__ out) {
    const int64_t groups = N >> 2;
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * blockDim.x) {
        float v[4 * M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float4 q = reinterpret_cast<const float4*>(s)[gi * M + j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
        float res[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[M];
#pragma unroll
            for (int m = 0; m < M; ++m) x[m] = v[i * M + m];
            res[i] = __fdiv_rn(np_sum_row<M>(x), (float)M);
        }
        reinterpret_cast<float4*>(out)[gi] = make_float4(res[0], res[1], res[2], res[3]);
    }
    // tail rows (N % 4)
    const int64_t t = (groups << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N && (int64_t)blockIdx.x * blockDim.x + threadIdx.x < 4) {
        float x[M];
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = s[t * M + m];
        out[t] = __fdiv_rn(np_sum_row<M>(x), (float)M);
    }
}

// mean over members, float32 (Ensemble default combine_with)
__global__ void k_ensemble_mean(const float* __restrict__ s, int64_t N, int M, float* __restrict__ out) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        auto ld = [&](int64_t i) { return s[i]; };
        float sum = np_pairwise<float>(ld, n * M, M);
        out[n] = __fdiv_rn(sum, (float)M);
    }
}

// AdaptiveEnsemble: np.sum(weights * scores, axis=1) -- float64 product then
// pairwise float64 sum (adaptive_ensemble.py:54,102)
__global__ void k_ensemble_wsum(const float* __restrict__ s, int64_t N, int M, const double* __restrict__ w,
                                double* __restrict__ out) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = n * M;
        auto ld = [&](int64_t i) { return __dmul_rn(w[i - b], (double)s[i]); };
        out[n] = np_pairwise<double>(ld, b, M);
    }
}

// ---- noisy_abstract_model.py:93-94:  alpha*signal + (1 - alpha)*noise, alpha = ss**d (host table)
__global__ void k_nam_combine(int64_t Q, const double* __restrict__ signal, const double* __restrict__ noise,
                              const int32_t* __restrict__ d, const double* __restrict__ alpha_tab, int n_tab,
                              double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Q; i += (int64_t)gridDim.x * blockDim.x) {
        int di = d[i];
        di = di < 0 ? 0 : (di >= n_tab ? n_tab - 1 : di);
        const double alpha = alpha_tab[di];
        // same operation order as the Python expression: (alpha*signal) + ((1-alpha)*noise)
        out[i] = __dadd_rn(__dmul_rn(alpha, signal[i]), __dmul_rn(1.0 - alpha, noise[i]));
    }
}

// ---- one_hot_to_string (sequence_utils.py:65-66): per-row np.argmax, first max wins, NaN = max
__global__ void k_argmax_decode(const double* __restrict__ x, int64_t rows, int A,
                                const uint8_t* __restrict__ alphabet, uint8_t* __restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const double* row = x + r * A;
        int best = 0;
        double bv = row[0];
        for (int a = 1; a < A; ++a) {
            const double v = row[a];
            if (v > bv || (v != v && bv == bv)) { best = a; bv = v; }
        }
        out[r] = alphabet[best];
    }
}

// ---- table landscape (tf_binding.py:43-44): fitness = table[packed k-mer]; unknown character -> NaN
__global__ void k_table_lookup(const double* __restrict__ table, int64_t len, const uint8_t* __restrict__ ascii,
                               const uint8_t* __restrict__ lut, int64_t N, int L, int bits, double* __restrict__ out) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        int64_t idx = 0;
        bool ok = true;
        for (int i = 0; i < L; ++i) {
            const int c = lut[ascii[n * L + i]];
            ok &= (c != 0xFF);
            idx |= (int64_t)(c & ((1 << bits) - 1)) << (bits * i);
        }
        out[n] = (ok && idx < len) ? table[idx] : __longlong_as_double(0x7ff8000000000000ll);
    }
}

// ---- additive landscape (additive_aav_packaging.py:101-107): out[n] = sum over positions, IN ORDER, of
// table[i][column(seq[n][i])] in float64 (one rounding per add, like the Python `+=` loop); residues a
// position has no entry for map to an all-zero column.  A workgroup stages S whole rows (S*L bytes, one
// contiguous 16-byte-aligned span) into LDS with coalesced loads; thread t then walks row t.
__global__ void __launch_bounds__(256) k_additive_sum(const double* __restrict__ table, int L, int ncol,
                                                      const uint8_t* __restrict__ ascii, const uint8_t* __restrict__ lut,
                                                      int64_t N, int S, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    __shared__ uint8_t slut[256];
    const int tid = threadIdx.x;
    slut[tid] = lut[tid];
    for (int64_t s0 = (int64_t)blockIdx.x * S; s0 < N; s0 += (int64_t)gridDim.x * S) {
        const int ns = (int)(N - s0 < S ? N - s0 : S);
        const int64_t nbytes = (int64_t)ns * L;
        const uint8_t* src = ascii + s0 * L;
        __syncthreads();
        const int64_t nvec = nbytes >> 4;
        for (int64_t v = tid; v < nvec; v += 256)
            reinterpret_cast<uint4*>(tile)[v] = reinterpret_cast<const uint4*>(src)[v];
        for (int64_t b = (nvec << 4) + tid; b < nbytes; b += 256) tile[b] = src[b];
        __syncthreads();
        if (tid < ns) {
            const uint8_t* row = tile + (size_t)tid * L;
            double acc = 0.0;
            for (int i = 0; i < L; ++i) acc = __dadd_rn(acc, table[(int64_t)i * ncol + slut[row[i]]]);
            out[s0 + tid] = acc;
        }
    }
}

// ---- test hook: ONE v_mfma_f32_16x16x4_f32 on caller-supplied per-lane operands, so the
// operand / result lane layout the scoring kernels rely on is checked on the real hardware.
__global__ void k_mfma_probe(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                             float* __restrict__ d) {
    const int lane = threadIdx.x;
    f4 acc = *reinterpret_cast<const f4*>(c + 4 * lane);
    acc = mfma16(a[lane], b[lane], acc);
    *reinterpret_cast<f4*>(d + 4 * lane) = acc;
}

inline unsigned grid_for(int64_t n, int block, int cus) {
    int64_t g = (n + block - 1) / block;
    int64_t cap = (int64_t)cus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

// ---- NoisyAbstractModel on a table landscape, everything behind the neighbour search in one kernel
// (noisy_abstract_model.py:86-94): signal = table[query], scale = table[nearest cached neighbour], noise = scale x E (what
// np.random.exponential(scale) returns for the standard-exponential draw E), out = alpha^d signal + (1 - alpha^d) noise --
// the look-up of k_table_lookup and the operation order of k_nam_combine.  flags: 1 = query or neighbour not in the table,
// 2 = negative neighbour value (the reference then draws from the cache instead: the caller redoes the batch).
__global__ void k_nam_table_blend(int64_t Q, const uint8_t* __restrict__ queries, const uint8_t* __restrict__ keys,
                                  const int64_t* __restrict__ arg, const int32_t* __restrict__ d,
                                  const double* __restrict__ table, int64_t len, const uint8_t* __restrict__ lut, int L, int bits,
                                  const double* __restrict__ E, const double* __restrict__ alpha_tab, int n_tab,
                                  double* __restrict__ out, int32_t* __restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Q; i += (int64_t)gridDim.x * blockDim.x) {
        auto look = [&](const uint8_t* row) {
            int64_t idx = 0;
            bool ok = true;
            for (int k = 0; k < L; ++k) {
                const int c = lut[row[k]];
                ok &= (c != 0xFF);
                idx |= (int64_t)(c & ((1 << bits) - 1)) << (bits * k);
            }
            return (ok && idx < len) ? table[idx] : __longlong_as_double(0x7ff8000000000000ll);
        };
        const int64_t j = arg[i];
        const double sig = look(queries + i * L);
        const double nbf = j >= 0 ? look(keys + j * L) : sig;          // (empty cache: the query is its own neighbour)
        int f = 0;
        if (sig != sig || nbf != nbf) f |= 1;
        if (nbf < 0.0) f |= 2;
        flags[i] = f;
        int di = d[i];
        di = di < 0 ? 0 : (di >= n_tab ? n_tab - 1 : di);
        const double alpha = alpha_tab[di];
        const double noise = __dmul_rn(nbf, E[i]);
        out[i] = __dadd_rn(__dmul_rn(alpha, sig), __dmul_rn(1.0 - alpha, noise));
    }
}

int fx_launch_nam_table_blend(fx_engine* e, int64_t Q, const uint8_t* d_queries, const uint8_t* d_keys, const int64_t* d_arg,
                              const int32_t* d_dist, const double* d_table, int64_t len, int L, int bits, const double* d_E,
                              const double* d_alpha, int n_tab, double* d_out, int32_t* d_flags) {
    if (Q == 0) return FX_OK;
    dim3 grid(grid_for(Q, 256, e->num_cus)), block(256);
    hipLaunchKernelGGL(k_nam_table_blend, grid, block, 0, e->stream, Q, d_queries, d_keys, d_arg, d_dist, d_table, len, e->d_lut, L, bits,
                       d_E, d_alpha, n_tab, d_out, d_flags);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_table_lookup(fx_engine* e, const double* d_table, int64_t len, const uint8_t* d_ascii, int64_t N,
                           int L, int bits, double* d_out) {
    if (N == 0) return FX_OK;
    dim3 grid(grid_for(N, 256, e->num_cus)), block(256);
    hipLaunchKernelGGL(k_table_lookup, grid, block, 0, e->stream, d_table, len, d_ascii, e->d_lut, N, L, bits, d_out);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_additive_sum(fx_engine* e, const double* d_table, int L, int ncol, const uint8_t* d_ascii, int64_t N,
                           double* d_out) {
    if (N == 0) return FX_OK;
    if (L < 1 || L > 3072) return fx_fail(e, FX_EUNSUPPORTED, "additive landscape: sequence length must be 1..3072");
    int S = (49152 / L) & ~15;                        // rows per tile: whole rows, 16-byte-aligned span, <= 48 KiB
    if (S > 256) S = 256;
    if (S < 16) S = 16;
    dim3 grid(grid_for(N, S, e->num_cus)), block(256);
    hipLaunchKernelGGL(k_additive_sum, grid, block, (size_t)S * L, e->stream, d_table, L, ncol, d_ascii, e->d_lut, N, S, d_out);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_mfma_probe(fx_engine* e, const float* d_a, const float* d_b, const float* d_c, float* d_d) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, e->stream, d_a, d_b, d_c, d_d);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_encode_onehot(fx_engine* e, const uint8_t* d_ascii, int64_t N, int L, int A, float* d_out) {
    const int64_t rows = N * L;
    if (rows == 0) return FX_OK;
    dim3 grid(grid_for(rows, 256, e->num_cus)), block(256);
    if (A == 4) {
        hipLaunchKernelGGL(k_encode_onehot<4>, grid, block, 0, e->stream, d_ascii, e->d_lut, rows, A, d_out, e->d_err);
    } else if (A == 20) {
        dim3 g2(grid_for(rows * 5, 256, e->num_cus));
        hipLaunchKernelGGL(k_encode_onehot<20>, g2, block, 0, e->stream, d_ascii, e->d_lut, rows, A, d_out, e->d_err);
    } else if (A % 4 == 0) {
        dim3 g2(grid_for(rows * (A / 4), 256, e->num_cus));
        hipLaunchKernelGGL(k_encode_onehot<1>, g2, block, 0, e->stream, d_ascii, e->d_lut, rows, A, d_out, e->d_err);
    } else {
        dim3 g2(grid_for(rows * A, 256, e->num_cus));
        hipLaunchKernelGGL(k_encode_onehot<0>, g2, block, 0, e->stream, d_ascii, e->d_lut, rows, A, d_out, e->d_err);
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

// Member-major planes (the engine's intermediate when only the mean is asked for): thread i reduces rows 4i..4i+3
// with one 16-byte load per member plane and one 16-byte store -- same NumPy summation order.
template <int M>
__global__ void k_ensemble_mean_planar(const float* __restrict__ planes, int64_t N, int64_t stride, float* __restrict__ out,
                                       bool out_aligned) {
    const int64_t groups = (N + 3) >> 2;
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * blockDim.x) {
        float4 q[M];
#pragma unroll
        for (int m = 0; m < M; ++m) q[m] = reinterpret_cast<const float4*>(planes + m * stride)[gi];
        float res[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[M];
#pragma unroll
            for (int m = 0; m < M; ++m) x[m] = i == 0 ? q[m].x : (i == 1 ? q[m].y : (i == 2 ? q[m].z : q[m].w));
            res[i] = __fdiv_rn(np_sum_row<M>(x), (float)M);
        }
        if (out_aligned && 4 * gi + 3 < N) reinterpret_cast<float4*>(out)[gi] = make_float4(res[0], res[1], res[2], res[3]);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * gi + i < N) out[4 * gi + i] = res[i];
        }
    }
}

int fx_launch_ensemble_mean_planar(fx_engine* e, const float* d_planes, int64_t N, int M, int64_t stride, float* d_out32) {
    if (N == 0) return FX_OK;
    dim3 gs(grid_for((N + 3) / 4, 256, e->num_cus)), block(256);
    const bool aligned = (reinterpret_cast<uintptr_t>(d_out32) & 15) == 0;   // (the planes always are)
    switch (M) {
#define FX_PLANAR_CASE(m) case m: hipLaunchKernelGGL(k_ensemble_mean_planar<m>, gs, block, 0, e->stream, d_planes, N, stride, d_out32, aligned); break;
        FX_PLANAR_CASE(1) FX_PLANAR_CASE(2) FX_PLANAR_CASE(3) FX_PLANAR_CASE(4) FX_PLANAR_CASE(5) FX_PLANAR_CASE(6)
        FX_PLANAR_CASE(7) FX_PLANAR_CASE(8) FX_PLANAR_CASE(9) FX_PLANAR_CASE(10) FX_PLANAR_CASE(11) FX_PLANAR_CASE(12)
        FX_PLANAR_CASE(13) FX_PLANAR_CASE(14) FX_PLANAR_CASE(15) FX_PLANAR_CASE(16)
#undef FX_PLANAR_CASE
        default: return fx_fail(e, FX_EUNSUPPORTED, "planar mean: more than 16 members");
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_ensemble_reduce(fx_engine* e, const float* d_scores, int64_t N, int M, const double* d_weights,
                              float* d_out32, double* d_out64) {
    if (N == 0) return FX_OK;
    dim3 grid(grid_for(N, 256, e->num_cus)), block(256);
    if (d_weights == nullptr) {
        dim3 gs(grid_for((N + 3) / 4, 256, e->num_cus));
        const bool aligned = ((reinterpret_cast<uintptr_t>(d_scores) | reinterpret_cast<uintptr_t>(d_out32)) & 15) == 0;
        switch (aligned ? M : 0) {
#define FX_MEAN_CASE(m) case m: hipLaunchKernelGGL(k_ensemble_mean_small<m>, gs, block, 0, e->stream, d_scores, N, d_out32); break;
            FX_MEAN_CASE(1) FX_MEAN_CASE(2) FX_MEAN_CASE(3) FX_MEAN_CASE(4) FX_MEAN_CASE(5) FX_MEAN_CASE(6)
            FX_MEAN_CASE(7) FX_MEAN_CASE(8) FX_MEAN_CASE(9) FX_MEAN_CASE(10) FX_MEAN_CASE(11) FX_MEAN_CASE(12)
            FX_MEAN_CASE(13) FX_MEAN_CASE(14) FX_MEAN_CASE(15) FX_MEAN_CASE(16)
#undef FX_MEAN_CASE
            default: hipLaunchKernelGGL(k_ensemble_mean, grid, block, 0, e->stream, d_scores, N, M, d_out32);
        }
    } else
        hipLaunchKernelGGL(k_ensemble_wsum, grid, block, 0, e->stream, d_scores, N, M, d_weights, d_out64);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_argmax_decode(fx_engine* e, const double* d_onehot, int64_t rows, int A, const uint8_t* d_alphabet,
                            uint8_t* d_out) {
    if (rows == 0) return FX_OK;
    dim3 grid(grid_for(rows, 256, e->num_cus)), block(256);
    hipLaunchKernelGGL(k_argmax_decode, grid, block, 0, e->stream, d_onehot, rows, A, d_alphabet, d_out);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_nam_combine(fx_engine* e, int64_t Q, const double* d_signal, const double* d_noise,
                          const int32_t* d_dist, const double* d_alpha, int n_tab, double* d_out) {
    if (Q == 0) return FX_OK;
    dim3 grid(grid_for(Q, 256, e->num_cus)), block(256);
    hipLaunchKernelGGL(k_nam_combine, grid, block, 0, e->stream, Q, d_signal, d_noise, d_dist, d_alpha, n_tab, d_out);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}
