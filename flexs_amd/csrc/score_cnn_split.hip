// Two-kernel CNN path for the shapes without a fused instantiation (kernel_size 2..7, any hidden width up to 256;
// num_filters <= 64 for 4-letter alphabets (kernel_size <= 5 beyond 32 filters), 17..32 for the protein alphabet -- cnn.py:10-21 leaves all of them to the
// caller).
//
// The fused kernel is a template over (alphabet, kernel size, channel tiles, hidden tiles, ...): covering every
// combination multiplies instantiations.  Here the two halves meet in HBM instead: the conv part
// (k_score_cnn_mfma<..., HEAD = false>, score_cnn_kernel.h) leaves each tile's max-pooled features -- 16 sequences x
// 16*FT channels, already in the accumulator layout the next MFMA wants as its B operand -- in a scratch buffer
// (64*FT bytes per sequence), and k_cnn_head runs Dense-relu-Dense-relu-Dense on them.  Instantiations are
// |kernel sizes| x |channel tiles| conv kernels + |channel tiles| x |hidden tile counts| head kernels.
#include "score_cnn_kernel.h"

namespace {

struct HeadArgs {
    const f4* pool;             // [(member * TG + tile) * FT + t][64 lanes]
    const float* w[FX_MAX_M];   // packed weights per member
    float* out;
    int64_t N, TG;
    int M, m_off;
    int64_t out_sn, out_sm;
    int rlh;
    int off_d1, off_d2, off_db, head_floats;   // the head's part of the packed image: [off_d1, off_d1 + head_floats)
};

// WLDS: the head's weights (FT*HT + HT*HT KiB) fit LDS; otherwise (13 / 16 hidden tiles) they stream from L2.
template <int FT, int HT, bool WLDS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_cnn_head(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, sq = lane & 15;
    int64_t u_lo, u_hi;
    fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();
        if (WLDS) fill_lds(reinterpret_cast<f4*>(smem), reinterpret_cast<const f4*>(p.w[m] + p.off_d1), p.head_floats / 4);
        __syncthreads();
        const float* base = WLDS ? smem : p.w[m] + p.off_d1;
        const f4* w_d1 = reinterpret_cast<const f4*>(base);
        const f4* w_d2 = reinterpret_cast<const f4*>(base + (p.off_d2 - p.off_d1));
        const float* db = base + (p.off_db - p.off_d1);
        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        for (int64_t tile = t_lo + wave; tile < t_hi; tile += WAVES) {
            asm volatile("" ::: "memory");               // keep the weight reads inside the tile loop
            if (!WLDS) asm volatile("" : "+v"(w_d1), "+v"(w_d2), "+v"(db));
            f4 pooled[FT][1];
            const f4* src = p.pool + (((int64_t)m * p.TG + tile) * FT) * 64 + lane;
#pragma unroll
            for (int t = 0; t < FT; ++t) pooled[t][0] = src[t * 64];
            f4 h1[HT][1], h2[HT][1];
            init_bias<HT, 1>(db, h1, g);
            mma_layer<FT, HT, 1>(w_d1, pooled, h1, lane);
            relu_tiles<HT, 1>(h1);
            init_bias<HT, 1>(db + 16 * HT, h2, g);
            mma_layer<HT, HT, 1>(w_d2, h1, h2, lane, p.rlh);
            relu_tiles<HT, 1>(h2);
            float y[1];
            final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2, y, g);
            const int64_t n = tile * 16 + sq;
            if (g == 0 && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[0]);
        }
    }
}

// Hidden sizes above 128 at batch size (round 6): the head's H x H layer (169 KiB at H = 200) streamed from L2 per tile and wave made the
// fused kernel L2-latency-bound (0.61 of the pipe at 1e5 sequences).  Here the WORKGROUP streams it once per lockstep round of WAVES tiles
// through two LDS slabs (mma_layer_slab_dma, the dense kernel's slab form), the first dense layer's blocks and the vectors are LDS-resident.
// A last round of at most three live tiles is walked without slabs (blocks straight from L2, no barriers) instead of costing a whole
// round.  Same MFMA sequence per output element as k_cnn_head and the fused kernel: the same bits.
template <int FT, int HT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_cnn_head_slab(HeadArgs p) {
    constexpr int KG = 2, LONE = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, sq = lane & 15;
    const int vec_floats = p.head_floats - (p.off_db - p.off_d1);
    float* vec_s = smem + FT * HT * 256;                                 // behind the first layer's blocks
    f4* slab = reinterpret_cast<f4*>(vec_s + ((vec_floats + 3) & ~3));
    int64_t u_lo, u_hi;
    fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();                                                 // the previous member's readers are done
        fill_lds(reinterpret_cast<f4*>(smem), reinterpret_cast<const f4*>(p.w[m] + p.off_d1), FT * HT * 64);
        for (int i = tid; i < vec_floats; i += WAVES * 64) vec_s[i] = p.w[m][p.off_db + i];
        __syncthreads();
        const f4* w_d1 = reinterpret_cast<const f4*>(smem);
        const f4* w_d2 = reinterpret_cast<const f4*>(p.w[m] + p.off_d2);
        const float* db = vec_s;
        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        FxSlabStream st{0, false};
        for (int64_t r0 = t_lo; r0 < t_hi; r0 += WAVES) {
            const int64_t want = r0 + wave;
            const bool live = want < t_hi;
            const int64_t tile = live ? want : t_lo;                     // (a wave without a tile runs along on the first one)
            const bool lone = r0 > t_lo && t_hi - r0 <= LONE;            // a last round of few tiles: no slabs
            const int64_t left = t_hi - (r0 + WAVES);                    // tiles behind this round
            const bool more = left > LONE;                               // ... and whether the next round streams slabs again
            asm volatile("" ::: "memory");
            asm volatile("" : "+v"(w_d2));
            if (lone && !live) continue;
            f4 pooled[FT][1];
            const f4* src = p.pool + (((int64_t)m * p.TG + tile) * FT) * 64 + lane;
#pragma unroll
            for (int t = 0; t < FT; ++t) pooled[t][0] = src[t * 64];
            f4 h1[HT][1], h2[HT][1];
            init_bias<HT, 1>(db, h1, g);
            mma_layer<FT, HT, 1>(w_d1, pooled, h1, lane);
            relu_tiles<HT, 1>(h1);
            init_bias<HT, 1>(db + 16 * HT, h2, g);
            if (lone) mma_layer<HT, HT, 1>(w_d2, h1, h2, lane, p.rlh);
            else mma_layer_slab_dma<HT, HT, KG, WAVES>(w_d2, more ? w_d2 : nullptr, slab, h1, h2, lane, p.rlh, st);
            relu_tiles<HT, 1>(h2);
            float y[1];
            final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2, y, g);
            const int64_t n = tile * 16 + sq;
            if (live && g == 0 && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[0]);
        }
    }
}

template <int FT, int HT>
int launch_head(fx_engine* e, const HeadArgs& a) {
    constexpr int WAVES = 8;
    if constexpr (HT >= 13) {
        // at least two lockstep rounds per workgroup: the slab form
        const int64_t U = (int64_t)a.M * a.TG;
        const size_t lds = (size_t)FT * HT * 1024 + (size_t)((a.head_floats - (a.off_db - a.off_d1) + 3) & ~3) * 4 + (size_t)2 * 2 * HT * 1024;
        if (e->cnn_head_slab && U >= (int64_t)e->num_cus * 2 * WAVES && lds <= (size_t)e->max_lds && a.off_d1 % 4 == 0) {
            auto kern = k_cnn_head_slab<FT, HT, WAVES>;
            static bool attr_set_s[64] = {};
            if (!attr_set_s[e->device & 63]) {
                FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set_s[e->device & 63] = true;
            }
            const int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds, e->stream, a);
            FX_HIP(e, hipGetLastError());
            return FX_OK;
        }
    }
    const size_t lds = (size_t)a.head_floats * 4 + 16;
    const bool wlds = lds <= (size_t)e->max_lds && HT <= 8;
    const int64_t U = (int64_t)a.M * a.TG;
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    if (blocks > (U + WAVES - 1) / WAVES) blocks = (U + WAVES - 1) / WAVES;
    if (blocks < 1) blocks = 1;
    if (wlds) {
        auto kern = k_cnn_head<FT, HT, true, WAVES>;
        static bool attr_set[64] = {};
        if (!attr_set[e->device & 63]) {
            FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[e->device & 63] = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds, e->stream, a);
    } else {
        hipLaunchKernelGGL((k_cnn_head<FT, HT, false, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), 16, e->stream, a);
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

template <int FT>
int dispatch_head(fx_engine* e, const HeadArgs& a, int HT) {
    switch (HT) {
        case 1: return launch_head<FT, 1>(e, a);
        case 2: return launch_head<FT, 2>(e, a);
        case 4: return launch_head<FT, 4>(e, a);
        case 7: return launch_head<FT, 7>(e, a);
        case 8: return launch_head<FT, 8>(e, a);
        case 13: return launch_head<FT, 13>(e, a);
        case 16: return launch_head<FT, 16>(e, a);
        default: return FX_EUNSUPPORTED;
    }
}

template <int K, int FT>
int launch_conv(fx_engine* e, const CnnArgs& a, size_t lds) {
    // conv part only: one tile per wave, 8 waves, conv weights in LDS, first layer in gather form
    if constexpr (K * 3 <= 16) {
        // small batch of long sequences (an explorer's 1-100 sequence call): the 8 waves of a workgroup split one
        // tile's positions, as in the fused kernel (score_cnn_mfma.hip) -- bit-identical to the one-wave walk
        const int L1 = a.L - K + 1;
        const bool seg = e->cnn_seg != 0 && (int64_t)a.M * a.TG <= e->num_cus && (e->cnn_seg > 0 || L1 >= 24) &&
                         lds + (size_t)8 * FT * 64 * 16 <= (size_t)e->max_lds;
        if (seg) return launch_g<4, K, FT, 1, 1, false, 8, true, 0, false, true, false>(e, a, lds);
    }
    if constexpr (K == 5 && FT == 2) {
        // batch launches (the wide-head path, round 6): four waves per SIMD, and the unrolled walk of the canonical short landscape
        if ((int64_t)a.M * a.TG >= (int64_t)e->num_cus * 16 && !e->rows_req.on) {
            if (a.L == 8) return launch_g<4, 5, 2, 1, 1, false, 16, true, 4, true, false, false>(e, a, lds);
            return launch_g<4, 5, 2, 1, 1, false, 16, true, 0, false, false, false>(e, a, lds);
        }
    }
    return launch_g<4, K, FT, 1, 1, false, 8, true, 0, false, false, false>(e, a, lds);
}

template <int FT>
int dispatch_conv(fx_engine* e, const CnnArgs& a, size_t lds, int K) {
    switch (K) {
        case 2: return launch_conv<2, FT>(e, a, lds);
        case 3: return launch_conv<3, FT>(e, a, lds);
        case 4: return launch_conv<4, FT>(e, a, lds);
        case 5: return launch_conv<5, FT>(e, a, lds);
        default: break;
    }
    if constexpr (FT <= 2) {                 // (3-4 channel tiles: the 6- and 7-tap windows do not fit the register file)
        if (K == 6) return launch_conv<6, FT>(e, a, lds);
        if (K == 7) return launch_conv<7, FT>(e, a, lds);
    }
    return FX_EUNSUPPORTED;
}

template <typename Args, typename Fn1, typename Fn2, typename Fn3, typename Fn4>
int by_channel_tiles(int FT, Fn1 f1, Fn2 f2, Fn3 f3, Fn4 f4_) {
    switch (FT) {
        case 1: return f1();
        case 2: return f2();
        case 3: return f3();
        case 4: return f4_();
        default: return FX_EUNSUPPORTED;
    }
}

}  // namespace

int fx_launch_score_cnn_split(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                              float* d_out_NM, int Mtot, int m_off) {
    if (N == 0) return FX_OK;
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    for (int m = 0; m < M; ++m) {
        const FxShape& t = models[m]->shape;
        if (t.kind != FX_CNN || t.L != s.L || t.A != s.A || t.F != s.F || t.H != s.H || t.K != s.K) return FX_EUNSUPPORTED;
    }
    // (binary alphabet, sequence_utils.py:16's BA: the canonical conv shape only -- its fused kernel covers 97..112 hidden units, this path the rest)
    if ((s.A != 4 && s.A != 20 && s.A != 2) || lay.FT < 1 || lay.FT > 4 || s.K < 2 || s.K > 7 || s.H > 256 || M > FX_MAX_M ||
        e->cnn_conv1_mfma || (s.A == 20 && (lay.FT != 2 || !e->cnn_pair)) || (s.A == 2 && (lay.FT != 2 || s.K != 5)))
        return FX_EUNSUPPORTED;
    const size_t conv_lds = (size_t)lay.conv_floats * 4 + 256 + 48;
    if (s.A != 20 && conv_lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    const int64_t TG = (N + 15) / 16;
    void* pool = nullptr;
    int rc = fx_scratch(e, 2, (size_t)M * (size_t)TG * lay.FT * 64 * sizeof(f4), &pool);
    if (rc) return rc;
    if (s.A == 20) {
        // protein alphabet: the conv part is the two-waves-per-tile kernel (score_cnn_pair.hip), same pool layout
        if ((rc = fx_launch_cnn_pair_conv(e, models, M, d_ascii, N, pool))) return rc;
        HeadArgs hp{};
        hp.pool = (const f4*)pool; hp.out = d_out_NM; hp.N = N; hp.TG = TG; hp.M = M; hp.m_off = m_off;
        for (int m = 0; m < M; ++m) hp.w[m] = models[m]->d_packed;
        hp.out_sn = e->planar_stride ? 1 : Mtot; hp.out_sm = e->planar_stride ? e->planar_stride : 1;
        hp.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
        hp.off_d1 = (int)lay.off_d1; hp.off_d2 = (int)lay.off_d2; hp.off_db = (int)lay.off_db;
        hp.head_floats = (int)(lay.total_floats - lay.off_d1);
        return dispatch_head<2>(e, hp, lay.HT);
    }

    CnnArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.N = N; a.TG = TG; a.M = M; a.Mtot = Mtot; a.m_off = m_off; a.L = s.L; a.rlh = 4;
    a.off_first = (int)lay.off_first; a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3;
    a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p; a.conv_floats = (int)lay.conv_floats; a.off_d1 = (int)lay.off_d1;
    a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db; a.total_floats = (int)lay.total_floats;
    a.pool_out = (f4*)pool;
    if (s.A == 2) rc = launch_g<2, 5, 2, 1, 1, false, 8, true, 0, false, false, false>(e, a, conv_lds);
    else rc = by_channel_tiles<CnnArgs>(lay.FT, [&] { return dispatch_conv<1>(e, a, conv_lds, s.K); }, [&] { return dispatch_conv<2>(e, a, conv_lds, s.K); },
                                        [&] { return dispatch_conv<3>(e, a, conv_lds, s.K); }, [&] { return dispatch_conv<4>(e, a, conv_lds, s.K); });
    if (rc) return rc;

    HeadArgs h{};
    h.pool = (const f4*)pool; h.out = d_out_NM; h.N = N; h.TG = TG; h.M = M; h.m_off = m_off;
    for (int m = 0; m < M; ++m) h.w[m] = models[m]->d_packed;
    h.out_sn = a.out_sn; h.out_sm = a.out_sm;
    h.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    h.off_d1 = (int)lay.off_d1; h.off_d2 = (int)lay.off_d2; h.off_db = (int)lay.off_db;
    h.head_floats = (int)(lay.total_floats - lay.off_d1);
    return by_channel_tiles<HeadArgs>(lay.FT, [&] { return dispatch_head<1>(e, h, lay.HT); }, [&] { return dispatch_head<2>(e, h, lay.HT); },
                                      [&] { return dispatch_head<3>(e, h, lay.HT); }, [&] { return dispatch_head<4>(e, h, lay.HT); });
}
