// Shape-agnostic scoring kernels (plain f32 VALU, one thread per sequence).
//
// These cover every (L, A, F, H, K) the reference constructors accept
// (cnn.py:10-21, mlp.py:10-19, global_epistasis_model.py:15-24), including the
// shapes the MFMA kernels do not instantiate (e.g. the reference's own smoke
// test: CNN(seq_len=3, num_filters=1, hidden_size=1, kernel_size=2),
// tests/test_models.py:55-62), and serve as an independent on-device
// cross-check of the MFMA kernels ("force_generic" engine option).
//
// Activations live in a global workspace laid out [row][thread] so that the
// 64 lanes of a wave touch consecutive floats; weights are read through
// wave-uniform (scalar) loads.
#include "fx_common.h"

namespace {

__device__ __forceinline__ float nan_to_num(float v) {
    // np.nan_to_num (keras_model.py:77): NaN -> 0, +-inf -> +-FLT_MAX
    if (v != v) return 0.f;
    if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return v;
}

struct GenericArgs {
    const uint8_t* ascii;
    const uint8_t* lut;
    const float* blob;      // Keras order
    float* ws;              // workspace: rows x T floats
    float* out;             // N x Mtot
    unsigned* err;
    int64_t N, n0;          // this launch covers sequences n0 .. n0 + T - 1
    int T;                  // threads in this launch (workspace pitch)
    int L, A, F, H, K;
    int Mtot, m;
    int64_t out_sn, out_sm; // out[n * out_sn + m * out_sm]
    int rows1;              // CNN: rows of the first workspace region = max(L1*F, H)
};

__device__ __forceinline__ int load_code(const GenericArgs& a, int64_t n, int l, bool& bad) {
    int c = a.lut[a.ascii[n * a.L + l]];
    if (c == 0xFF) { bad = true; c = 0; }
    return c;
}

// The two contraction helpers compute OB outputs per pass over the inputs: one workspace load (the lane's
// activation) then feeds OB fused multiply-adds whose weights are wave-uniform (scalar loads of OB consecutive
// floats), instead of one memory instruction per multiply-add.  Every output still accumulates its terms in the
// same order (bias first, inputs ascending), so results do not depend on OB.
constexpr int OB = 16;

// conv over workspace rows.  in rows [Lin*Cin], out rows [Lout*Cout]
__device__ void conv_ws(const float* in, float* out, int T, int Lin, int Cin, const float* __restrict__ w,
                        const float* __restrict__ b, int k, int Cout, int pl) {
    for (int t = 0; t < Lin; ++t)
        for (int o0 = 0; o0 < Cout; o0 += OB) {
            float acc[OB];
#pragma unroll
            for (int u = 0; u < OB; ++u) acc[u] = o0 + u < Cout ? b[o0 + u] : 0.f;
            for (int j = 0; j < k; ++j) {
                const int p = t + j - pl;
                if (p < 0 || p >= Lin) continue;
                for (int c = 0; c < Cin; ++c) {
                    const float x = in[(int64_t)(p * Cin + c) * T];
                    const float* wr = w + ((int64_t)j * Cin + c) * Cout + o0;
#pragma unroll
                    for (int u = 0; u < OB; ++u)
                        if (o0 + u < Cout) acc[u] = fmaf(x, wr[u], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < OB; ++u)
                if (o0 + u < Cout) out[(int64_t)(t * Cout + o0 + u) * T] = fmaxf(acc[u], 0.f);
        }
}

__device__ void dense_ws(const float* in, float* out, int T, int nin, const float* __restrict__ w,
                         const float* __restrict__ b, int nout) {
    for (int o0 = 0; o0 < nout; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int u = 0; u < OB; ++u) acc[u] = o0 + u < nout ? b[o0 + u] : 0.f;
        for (int i = 0; i < nin; ++i) {
            const float x = in[(int64_t)i * T];
            const float* wr = w + (int64_t)i * nout + o0;
#pragma unroll
            for (int u = 0; u < OB; ++u)
                if (o0 + u < nout) acc[u] = fmaf(x, wr[u], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < OB; ++u)
            if (o0 + u < nout) out[(int64_t)(o0 + u) * T] = fmaxf(acc[u], 0.f);
    }
}

__device__ float dot_ws(const float* in, int T, int nin, const float* __restrict__ w, float b) {
    float acc = b;
    for (int i = 0; i < nin; ++i) acc = fmaf(in[(int64_t)i * T], w[i], acc);
    return acc;
}

__global__ void k_score_generic_cnn(GenericArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = a.n0 + tid;
    if (tid >= a.T || n >= a.N) return;
    const int L = a.L, A = a.A, F = a.F, H = a.H, K = a.K, K3 = A - 1, L1 = L - K + 1, T = a.T;
    const float* w1 = a.blob;             const float* b1 = w1 + (int64_t)K * A * F;
    const float* w2 = b1 + F;             const float* b2 = w2 + (int64_t)K * F * F;
    const float* w3 = b2 + F;             const float* b3 = w3 + (int64_t)K3 * F * F;
    const float* d1 = b3 + F;             const float* c1 = d1 + (int64_t)F * H;
    const float* d2 = c1 + H;             const float* c2 = d2 + (int64_t)H * H;
    const float* d3 = c2 + H;             const float* c3 = d3 + H;
    float* buf1 = a.ws + tid;                         // L1*F rows
    float* buf2 = buf1 + (int64_t)a.rows1 * T;        // max(L1*F, F+H) rows
    bool bad = false;
    // conv1 (valid) on the one-hot input == gather of kernel rows (sequence_utils.py:44-47)
    for (int t = 0; t < L1; ++t) {
        for (int f = 0; f < F; ++f) buf1[(int64_t)(t * F + f) * T] = b1[f];
        for (int j = 0; j < K; ++j) {
            const int c = load_code(a, n, t + j, bad);
            const float* row = w1 + ((int64_t)j * A + c) * F;
            for (int f = 0; f < F; ++f) buf1[(int64_t)(t * F + f) * T] += row[f];
        }
        for (int f = 0; f < F; ++f) {
            float& v = buf1[(int64_t)(t * F + f) * T];
            v = fmaxf(v, 0.f);
        }
    }
    conv_ws(buf1, buf2, T, L1, F, w2, b2, K, F, (K - 1) / 2);       // same
    conv_ws(buf2, buf1, T, L1, F, w3, b3, K3, F, (K3 - 1) / 2);     // same (MaxPooling1D(1) = id)
    float* pooled = buf2;                                           // F rows
    for (int f = 0; f < F; ++f) {                                   // GlobalMaxPooling1D
        float m = buf1[(int64_t)f * T];
        for (int t = 1; t < L1; ++t) m = fmaxf(m, buf1[(int64_t)(t * F + f) * T]);
        pooled[(int64_t)f * T] = m;
    }
    float* g1 = buf1;                                               // H rows
    float* g2 = buf2 + (int64_t)F * T;                              // H rows
    dense_ws(pooled, g1, T, F, d1, c1, H);
    dense_ws(g1, g2, T, H, d2, c2, H);
    float y = dot_ws(g2, T, H, d3, c3[0]);
    a.out[n * a.out_sn + a.m * a.out_sm] = nan_to_num(y);
    if (bad) fx_raise(a.err, FX_ERR_BADCHAR);
}

__global__ void k_score_generic_mlp(GenericArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = a.n0 + tid;
    if (tid >= a.T || n >= a.N) return;
    const int L = a.L, A = a.A, H = a.H, T = a.T;
    const float* d1 = a.blob;             const float* c1 = d1 + (int64_t)L * A * H;
    const float* d2 = c1 + H;             const float* c2 = d2 + (int64_t)H * H;
    const float* d3 = c2 + H;             const float* c3 = d3 + (int64_t)H * H;
    const float* d4 = c3 + H;             const float* c4 = d4 + H;
    float* g0 = a.ws + tid;
    float* g1 = g0 + (int64_t)H * T;
    bool bad = false;
    for (int o = 0; o < H; ++o) g0[(int64_t)o * T] = c1[o];
    for (int l = 0; l < L; ++l) {                                   // Flatten index l*A + a (mlp.py:23)
        const int c = load_code(a, n, l, bad);
        const float* row = d1 + ((int64_t)l * A + c) * H;
        for (int o = 0; o < H; ++o) g0[(int64_t)o * T] += row[o];
    }
    for (int o = 0; o < H; ++o) { float& v = g0[(int64_t)o * T]; v = fmaxf(v, 0.f); }
    dense_ws(g0, g1, T, H, d2, c2, H);
    dense_ws(g1, g0, T, H, d3, c3, H);
    float y = dot_ws(g0, T, H, d4, c4[0]);
    a.out[n * a.out_sn + a.m * a.out_sm] = nan_to_num(y);
    if (bad) fx_raise(a.err, FX_ERR_BADCHAR);
}

__global__ void k_score_generic_ge(GenericArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = a.n0 + tid;
    if (tid >= a.T || n >= a.N) return;
    const int L = a.L, A = a.A, H = a.H, T = a.T;
    const float* d1 = a.blob;             const float* c1 = d1 + (int64_t)L * A;
    const float* d2 = c1 + 1;             const float* c2 = d2 + H;
    const float* d3 = c2 + H;             const float* c3 = d3 + (int64_t)H * H;
    const float* d4 = c3 + H;             const float* c4 = d4 + H;
    float* g0 = a.ws + tid;
    float* g1 = g0 + (int64_t)H * T;
    bool bad = false;
    float s = c1[0];
    for (int l = 0; l < L; ++l) s += d1[(int64_t)l * A + load_code(a, n, l, bad)];
    s = fmaxf(s, 0.f);
    for (int o = 0; o < H; ++o) g0[(int64_t)o * T] = fmaxf(fmaf(s, d2[o], c2[o]), 0.f);
    dense_ws(g0, g1, T, H, d3, c3, H);
    float y = dot_ws(g1, T, H, d4, c4[0]);
    a.out[n * a.out_sn + a.m * a.out_sm] = nan_to_num(y);
    if (bad) fx_raise(a.err, FX_ERR_BADCHAR);
}

}  // namespace

int fx_launch_score_generic(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                            int64_t N, float* d_out_NM, int Mtot, int m_off) {
    if (N == 0) return FX_OK;
    for (int m = 0; m < M; ++m) {
        const FxShape& s = models[m]->shape;
        int64_t rows, rows1 = 0;
        if (s.kind == FX_CNN) {
            // region 1: conv buffers / g1 (H rows); region 2: conv buffer / pooled (F) + g2 (H)
            rows1 = (int64_t)s.L1() * s.F;
            if (rows1 < s.H) rows1 = s.H;
            int64_t rows2 = (int64_t)s.L1() * s.F;
            if (rows2 < s.F + s.H) rows2 = s.F + s.H;
            rows = rows1 + rows2;
        } else {
            rows = 2 * (int64_t)s.H;
        }
        // threads per launch: keep the workspace under ~1 GiB
        int64_t T = 65536;
        while (T > 256 && rows * T * 4 > ((int64_t)1 << 30)) T >>= 1;
        if (T > N) T = (N + 255) / 256 * 256;
        void* ws = nullptr;
        int rc = fx_scratch(e, 2, (size_t)(rows * T * 4), &ws);
        if (rc) return rc;
        for (int64_t n0 = 0; n0 < N; n0 += T) {
            GenericArgs a;
            a.ascii = d_ascii; a.lut = e->d_lut; a.blob = models[m]->d_blob; a.ws = (float*)ws;
            a.out = d_out_NM; a.err = e->d_err; a.N = N; a.n0 = n0; a.T = (int)T;
            a.L = s.L; a.A = s.A; a.F = s.F; a.H = s.H; a.K = s.K; a.Mtot = Mtot; a.m = m_off + m;
            a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
            a.rows1 = (int)rows1;
            dim3 grid((unsigned)((T + 255) / 256)), block(256);
            if (s.kind == FX_CNN) {
                hipLaunchKernelGGL(k_score_generic_cnn, grid, block, 0, e->stream, a);
            } else if (s.kind == FX_MLP) {
                hipLaunchKernelGGL(k_score_generic_mlp, grid, block, 0, e->stream, a);
            } else {
                hipLaunchKernelGGL(k_score_generic_ge, grid, block, 0, e->stream, a);
            }
            FX_HIP(e, hipGetLastError());
        }
    }
    return FX_OK;
}
