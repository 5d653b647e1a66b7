// K2, small launches of the MLP / GlobalEpistasis model: ONE tile of 16 sequences per workgroup, its OUTPUT tiles dealt
// to the 8 waves.
//
// An explorer-size call (1-100 sequences; DyNA-PPO's environment steps score 1-10 at a time with an ensemble that holds
// an MLP(200), flexs/baselines/explorers/dyna_ppo.py:53-55) is one tile per member.  The persistent kernel
// (score_dense_mfma.hip) then spends ~5 us copying a 150 KiB weight image into LDS for a single tile, or -- hidden sizes
// above 128, whose HxH blocks do not fit -- streams 2 x 169 KiB through LDS slabs behind barriers (51 us), and the first
// layer of a long sequence is a serial chain of row gathers (87 us at seq_len 100).  Here nothing is staged: wave w owns
// output tiles {w, w + 8} of every layer and reads exactly the weights of those tiles from global memory (L2) into
// registers -- an eighth of every matrix per wave, all loads of a layer in flight at once -- and the layers' outputs meet in
// 2 x HT KiB of LDS.  Every output element sees the arithmetic of the persistent kernel (first layer: bias + the same
// rows -- pre-summed pair rows where that kernel uses them -- in position order; hidden layers: (input tile, k-step)
// order with the same tail skip; the same final dot), so the scores are the SAME BITS (tested).  GlobalEpistasis: every
// wave sums the scalar first layer itself (lane group g takes positions g, g + 4, ... in order, two cross-lane adds --
// the persistent kernel's order), then owns its output tiles of the 1 -> H layer and of the H x H layer.
#include "fx_common.h"
#include "mfma_common.h"
#include "score_dense_tile.h"

namespace {

constexpr int SW = 8;                                     // waves per workgroup

struct SmallArgs {
    const uint8_t* ascii;
    const uint8_t* lut;
    const float* w[FX_MAX_M];
    float* out;
    unsigned* err;
    int64_t N, TG;
    int M, m_off;
    int64_t out_sn, out_sm;
    int L, A, rlh;
    int pair;                    // 1 = first layer from the pre-summed pair rows (4-letter alphabets), as the PAIR form of the persistent kernel
    int off_w1p, off_w1pair, off_d2, off_d3, off_db, off_first;
    // SERVER (the resident form, see score_cnn_quad.hip): workgroup = (member, tile slot), requests from the mailboxes
    int srv_tiles; int srv_fast; int srv_sleep; int srv_fence; FxMailIn* min; FxMailOut* mout;
    unsigned long long idle_ticks, life_ticks;
};

template <int KIND, int HT, bool SERVER = false>
__global__ void __launch_bounds__(SW * 64) k_score_dense_small(SmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L;
    f4* hx = reinterpret_cast<f4*>(smem);                 // [2][HT][64]: layer outputs, double-buffered
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem + 2 * HT * 256);
    uint8_t* bytes_s = lut_s + 256;                       // the tile's 16 x L bytes

    const int64_t unit = blockIdx.x;
    const int m = SERVER ? (int)(unit / p.srv_tiles) : (int)(unit / p.TG);
    const int64_t tg0 = SERVER ? unit % p.srv_tiles : unit - (int64_t)m * p.TG;    // the (first) tile of this workgroup
    int64_t Ncur = p.N;                                  // SERVER: the current request's batch
    // SERVER: request word, exit flag, bad-character flag behind the tile's bytes
    uint8_t* srv_area = bytes_s + ((16 * L + 15) & ~15);
    unsigned long long& srv_req = *reinterpret_cast<unsigned long long*>(srv_area);
    int& srv_exit = *reinterpret_cast<int*>(srv_area + 8);
    volatile int& srv_bad = *reinterpret_cast<volatile int*>(srv_area + 12);
    volatile int& srv_abandon = *reinterpret_cast<volatile int*>(srv_area + 16);     // a streamed request whose rows never came: not answered
    [[maybe_unused]] unsigned long long srv_last = 0, srv_start = 0, srv_seen = 0;
    for (int i = tid; i < 64; i += SW * 64) reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];
    if constexpr (SERVER) {
        __syncthreads();
        if (wave == 0) srv_start = srv_seen = wall_clock64();          // (every lane of wave 0 keeps the wait's clocks: fx_server_wait_line)
        if (tid == 0)
            __hip_atomic_store(const_cast<unsigned*>(&p.mout->alive[p.m_off + m][tg0]), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  for (;;) {                                               // (SERVER: one iteration per request)
    if constexpr (SERVER) {
        if (wave == 0) {
            int ex = 0;
            unsigned long long r = 0;
            if (tg0 == 0) {
                // the slot of tile 0 polls the whole request line: a tiny request's bytes arrive with the word (FxMailIn::tiny)
                unsigned payload = 0;
                r = fx_server_wait_line(p.min, srv_last, srv_seen, srv_start, p.idle_ticks, p.life_ticks, lane, &ex, &payload);
                // (only the dwords that hold the request's N x L bytes -- N <= 16, one tile: the tile's byte rows are 16 x L bytes)
                if (!ex && (r & FX_SERVE_TINY) && lane >= 2 && lane < 14 && (lane - 2) * 4 < (int)(r & 0x3FFFull) * L)
                    reinterpret_cast<unsigned*>(bytes_s)[lane - 2] = payload;
            } else if (lane == 0) {
                r = fx_server_wait(p.min, srv_last, srv_seen, srv_start, p.idle_ticks, p.life_ticks, (int)tg0 < p.srv_fast, p.srv_sleep, &ex);
            }
            if (lane == 0) { srv_req = r; srv_exit = ex; srv_bad = 0; srv_abandon = 0; }
        }
        __syncthreads();
        if (srv_exit) break;
        Ncur = (int64_t)(srv_req & 0x3FFFull);
        if (tg0 * 16 >= Ncur) {                              // a request with fewer tiles: nothing to answer from this slot
            if (wave == 0) { srv_last = srv_req; srv_seen = wall_clock64(); }
            __syncthreads();                                 // (everybody has read the request word)
            continue;
        }
    }
    bool bad = false;
    // SERVER: this slot's tiles of the request -- slot, slot + T, slot + 2 T, ... (T = tile slots per member); else the one tile
    for (int64_t tg = tg0; tg == tg0 || (SERVER && tg * 16 < Ncur); tg += SERVER ? p.srv_tiles : 1) {
        const int64_t rows = Ncur - tg * 16 < 16 ? Ncur - tg * 16 : 16;
        const int64_t n = tg * 16 + sq;
        if (SERVER && (srv_req & FX_SERVE_TINY)) {
            // (a tiny request: its bytes came with the request word, wave 0 has put them in place)
        } else if constexpr (SERVER) {
            // the tile's bytes, dword-wise and past the caches (the host wrote them through the BAR)
            const unsigned* src = reinterpret_cast<const unsigned*>(p.min->bytes + tg * 16 * L);
            // (a streamed request: the host is still packing; one that was given up is not waited for again)
            if (!srv_abandon && !fx_server_rows_ready(p.min, srv_req, tg * 16 + rows)) srv_abandon = 1;
            for (int i = tid; i * 4 < (int)rows * L; i += SW * 64)
                reinterpret_cast<unsigned*>(bytes_s)[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int i = tid; i < (int)rows * L; i += SW * 64) bytes_s[i] = p.ascii[tg * 16 * L + i];
        }
        __syncthreads();
        const uint8_t* row = bytes_s + (n < Ncur ? sq : 0) * L;    // lanes past the batch recompute the tile's first sequence
        const float* W = p.w[m];
        float y0 = 0.f;
        fx_dense_tile8<KIND, HT, false>(true, wave, lane, row, L, p.A, p.rlh, p.pair, W + p.off_first, W + p.off_w1p, W + p.off_w1pair,
                                        reinterpret_cast<const f4*>(W + p.off_d2), reinterpret_cast<const f4*>(W + p.off_d3), W + p.off_db,
                                        lut_s, hx, SERVER ? &srv_bad : nullptr, bad, y0);

        if (wave == 0) {
            float y[1] = {y0};
            if constexpr (SERVER) {
                // (score, tag) in one 8-byte SYSTEM-scope store to host memory (written through by itself; serve_fence = 1 adds round 3's fence)
                const unsigned tag = (unsigned)(srv_req >> 16) | (srv_bad ? 0x80000000u : 0u);
                if (g == 0 && n < Ncur && !srv_abandon)
                    __hip_atomic_store(const_cast<unsigned long long*>(&p.mout->ans[p.m_off + m][n]),
                                       ((unsigned long long)tag << 32) | __float_as_uint(fx_nan_to_num(y[0])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p.srv_fence) __threadfence_system();
            } else {
                if (g == 0 && n < Ncur) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[0]);
            }
        }
        if constexpr (SERVER) __syncthreads();               // (the byte rows and the exchange buffers are free for the slot's next tile)
    }
    if constexpr (SERVER) {
        if (wave == 0) { srv_last = srv_req; srv_seen = wall_clock64(); }
        __syncthreads();                                     // (the request word is free again)
    } else {
        if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
        break;
    }
  }
    if constexpr (SERVER) {
        if (tid == 0) __hip_atomic_store(const_cast<unsigned*>(&p.mout->alive[p.m_off + m][tg0]), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int KIND, int HT>
int launch_small(fx_engine* e, const SmallArgs& a, int64_t U) {
    auto kern = k_score_dense_small<KIND, HT>;
    const size_t lds = (size_t)2 * HT * 1024 + 256 + (size_t)16 * a.L;
    if (lds > 64 * 1024) return FX_EUNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)U), dim3(SW * 64), lds, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

template <int KIND, int HT>
int launch_small_server(fx_engine* e, const SmallArgs& a, int M, hipStream_t stream) {
    auto kern = k_score_dense_small<KIND, HT, true>;
    const size_t lds = (size_t)2 * HT * 1024 + 256 + (((size_t)16 * a.L + 15) & ~(size_t)15) + 32;
    if (lds > 64 * 1024) return FX_EUNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)(M * a.srv_tiles)), dim3(SW * 64), lds, stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

}  // namespace

// FX_EUNSUPPORTED when the small-launch form does not apply (the caller carries on with the persistent kernel).
int fx_launch_score_mlp_small(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                              float* d_out_NM, int Mtot, int m_off) {
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    if (!e->dense_small || (s.kind != FX_MLP && s.kind != FX_GE) || M > FX_MAX_M || s.A > 127) return FX_EUNSUPPORTED;
    const int64_t TG = (N + 15) / 16, U = (int64_t)M * TG;
    // One workgroup per CU is ~8-9 us for the canonical shapes against ~14 us of the persistent kernel (LDS fill + a lone tile).
    // Where that kernel's lone tile is a latency chain -- hidden layers streamed through LDS slabs (H > 128), a first
    // layer gathered row by row from L2 (long sequences / wide alphabets) -- this form wins up to ~4 workgroups per CU
    // (tools/archive/runs/r2_mlp_small_crossover.py); beyond that its per-tile re-read of the weights from L2 loses.
    int64_t per_cu = (lay.HT > 8 || (s.kind == FX_MLP && (int64_t)s.L * s.A >= 160)) ? 4 : 1;
    // (an MLP whose first-layer rows do not fit LDS -- protein alphabets -- has the position-major first layer from 2 tiles per CU on, round 6:
    //  16 384 sequences of L = 90, H = 200 take this form 105 us, 30 000 take that one 106, profiles/r6_protein_mlp_wide.log)
    if (fx_mlp_l1_pos_applies(e, s, lay) && !e->rows_req.on && !e->ascii_host) per_cu = e->mlp_l1_pos_tiles < per_cu ? e->mlp_l1_pos_tiles : per_cu;
    if (e->dense_small < 2 && U > per_cu * e->num_cus) return FX_EUNSUPPORTED;
    const int form = s.kind == FX_MLP ? fx_mlp_first_layer_form(e, s, lay) : 0;
    if (form > 1) return FX_EUNSUPPORTED;
    SmallArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.N = N; a.TG = TG; a.M = M; a.m_off = m_off;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.L = s.L; a.A = s.A; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.pair = form;
    a.off_w1p = (int)lay.off_w1p; a.off_w1pair = (int)lay.off_w1pair; a.off_d2 = (int)lay.off_d2; a.off_d3 = (int)lay.off_d3;
    a.off_db = (int)lay.off_db; a.off_first = (int)lay.off_first;
#define FX_SMALL_CASES(KIND)                                        \
    switch (lay.HT) {                                               \
        case 1: return launch_small<KIND, 1>(e, a, U);              \
        case 2: return launch_small<KIND, 2>(e, a, U);              \
        case 4: return launch_small<KIND, 4>(e, a, U);              \
        case 7: return launch_small<KIND, 7>(e, a, U);              \
        case 8: return launch_small<KIND, 8>(e, a, U);              \
        case 13: return launch_small<KIND, 13>(e, a, U);            \
        case 16: return launch_small<KIND, 16>(e, a, U);            \
        default: return FX_EUNSUPPORTED;                            \
    }
    if (s.kind == FX_GE) { FX_SMALL_CASES(FX_GE) }
    FX_SMALL_CASES(FX_MLP)
#undef FX_SMALL_CASES
}


// The resident form (see fx_launch_score_cnn_quad_server): one workgroup per member and tile slot, weights read from L2 at
// every request as the launched form does (nothing to fill), so a request costs the mailbox round trip + the tile.
int fx_launch_score_dense_small_server(fx_engine* e, fx_model* const* models, int M, int m_off, int tiles, hipStream_t stream,
                                       FxMailIn* d_in, FxMailOut* d_out, unsigned long long idle_ticks, unsigned long long life_ticks) {
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    if (!e->dense_small || (s.kind != FX_MLP && s.kind != FX_GE) || M > FX_MAX_M || M < 1 || s.A > 127) return FX_EUNSUPPORTED;
    for (int m = 1; m < M; ++m) {
        const FxShape& t = models[m]->shape;
        if (t.kind != s.kind || t.L != s.L || t.A != s.A || t.H != s.H) return FX_EUNSUPPORTED;
    }
    const int form = s.kind == FX_MLP ? fx_mlp_first_layer_form(e, s, lay) : 0;
    if (form > 1) return FX_EUNSUPPORTED;
    SmallArgs a{};
    a.lut = e->d_lut; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.M = M; a.m_off = m_off;
    a.L = s.L; a.A = s.A; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.pair = form;
    a.off_w1p = (int)lay.off_w1p; a.off_w1pair = (int)lay.off_w1pair; a.off_d2 = (int)lay.off_d2; a.off_d3 = (int)lay.off_d3;
    a.off_db = (int)lay.off_db; a.off_first = (int)lay.off_first;
    a.srv_tiles = tiles; a.srv_fast = e->server.fast; a.srv_sleep = (int)e->serve_poll_sleep; a.srv_fence = (int)e->serve_fence; a.min = d_in; a.mout = d_out; a.idle_ticks = idle_ticks; a.life_ticks = life_ticks;
#define FX_SMALL_CASES(KIND)                                               \
    switch (lay.HT) {                                                      \
        case 1: return launch_small_server<KIND, 1>(e, a, M, stream);      \
        case 2: return launch_small_server<KIND, 2>(e, a, M, stream);      \
        case 4: return launch_small_server<KIND, 4>(e, a, M, stream);      \
        case 7: return launch_small_server<KIND, 7>(e, a, M, stream);      \
        case 8: return launch_small_server<KIND, 8>(e, a, M, stream);      \
        case 13: return launch_small_server<KIND, 13>(e, a, M, stream);    \
        case 16: return launch_small_server<KIND, 16>(e, a, M, stream);    \
        default: return FX_EUNSUPPORTED;                                   \
    }
    if (s.kind == FX_GE) { FX_SMALL_CASES(FX_GE) }
    FX_SMALL_CASES(FX_MLP)
#undef FX_SMALL_CASES
}
