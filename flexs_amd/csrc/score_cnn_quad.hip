// K1, small launches: one tile of 16 sequences shared by FOUR waves (one per SIMD of the CU).
//
// With at most three tiles per workgroup (10^4 sequences on one member: 625 tiles for 256 CUs, configs[0]; every
// explorer-size call) the one-wave-per-tile kernel runs a LONE wave per SIMD: ~75 % of the pipe for that wave, three of
// the CU's four matrix pipes idle (profiles/archive/r2_trace_probe: 12 us per tile whose MFMAs are 8 us of one pipe).  Here a
// wave quad walks the network layer by layer and exchanges activations through LDS:
//     wave q takes the conv positions q, q + 4, q + 8, ... (TF-binding: seq_len 8, kernel 5 -> 4 positions, one each;
//     RNA: seq_len 14 -> 10 positions, 3-3-2-2; up to 12 positions = seq_len 16)
//     A  conv1 at its positions (row gather)                    -> X[pos]          barrier
//     B  conv2 at its positions from X[pos-2 .. pos+2]           -> Y[pos]          barrier
//     C  conv3 at its positions from Y[pos-1 .. pos+1], relu     -> X[pos]          barrier
//     D  global max over all positions; dense 1, output tiles {q, q+4}    -> Y      barrier
//     E  dense 2, output tiles {q, q+4} from all 7 tiles of Y    -> X               barrier
//     F  wave 0: final dot over all 7 tiles of X, nan_to_num, store
// Every output element sees exactly the MFMA / add sequence of the one-wave kernel (k-order (tap, input tile, k-step),
// padding taps skipped, hidden tail k-steps skipped), the exchanges are copies: results are BIT-IDENTICAL to
// k_score_cnn_mfma (tested), so a sequence scores the same in a call of 20 and in a batch of 10^5.
// Three quads per workgroup at 4 positions (X / Y: 2 x 8 KiB per quad next to the member's ~103 KiB of weights), one quad
// (X / Y: 2 x 24 KiB) at up to 12; X and Y swap roles every round so that wave 0's phase F never races the next round's
// phase A.
// Start-up: the LUT, the first round's 16 x 8 sequence bytes and the whole weight image are requested at once by direct
// global -> LDS copies (mfma_common.h fx_dma_fill); phase A starts when the conv part (~35 KiB) has landed, the dense
// head's ~68 KiB arrive during the convolutions and are waited for before phase D (engine option dma_fill).
#include "fx_common.h"
#include "mfma_common.h"
#include "np_sum.h"

namespace {


struct QuadArgs {
    const uint8_t* ascii;
    const uint8_t* lut;
    const float* w[FX_MAX_M];
    float* out;
    unsigned* err;
    unsigned long long* trace;  // in-kernel timeline of trace builds (null = off), see fx_stamp
    int64_t N, TG;
    int M, m_off;
    int64_t out_sn, out_sm;
    int L, rlh;
    int dma;                    // 1 = weights (and the first round's bytes) go global -> LDS directly, head part lands during the convolutions
    // fused ensemble mean (explorer-size calls): the member whose workgroup finishes a tile LAST averages the tile's 16 sequences
    float* mean_out;            // null = off
    unsigned* tickets;          // one per tile, zero between launches (fx_zero_pool; the last arrival resets its entry)
    // resident form
    int rotate; int srv_tiles; int srv_fast; int srv_sleep; int srv_fence; FxMailIn* min; FxMailOut* mout;               // null = ordinary launch
    unsigned long long idle_ticks, life_ticks;   // leave after this long without a request / in total (100 MHz ticks)
    int off_c2, off_c3, off_cb, off_w1p, off_d1, off_d2, off_db, total_floats;
};

// QUADS quads per workgroup; XT = 16-channel tiles per exchange buffer (>= FT x positions and >= HT); L1C > 0: the number
// of conv positions is a compile-time constant (seq_len 8: one position per wave, the position loops fold away)
// SERVER: the resident form (QUADS = 1).  One workgroup per (member, 16-sequence tile slot) loads its member's weights ONCE
// and then answers requests until told to stop, idle for idle_ticks, or life_ticks old: it polls the request word in
// DEVICE memory (FxMailIn: the host stores it through the BAR), stages its tile's bytes, runs the round below and stores
// (score, request tag) pairs + a system fence into pinned HOST memory (FxMailOut) -- an explorer-size call then costs a
// mailbox round trip (~3 us for a handful of workgroups, tools/probes/mailbox_probe2.hip) plus one round instead of a
// launch + a 3.5 us weight fill + a second launch for the mean + a completion wait (~27 us).  Same round code, same bits.
template <int HT, int QUADS, int XT, int L1C, int K = 5, bool SERVER = false>
__global__ void __launch_bounds__(QUADS * 256) k_score_cnn_quad(QuadArgs p) {
    constexpr int A = 4, K3 = 3, FT = 2, PL2 = (K - 1) / 2, PL3 = 1, QWAVES = 4 * QUADS;
    static_assert(K % 2 == 1, "'same' padding of conv2: (K - 1) / 2 on either side");
    static_assert(XT >= HT + 1, "the dense layers exchange HT tiles through the same buffers");
    const int L1 = L1C > 0 ? L1C : p.L - K + 1;           // (L1 * FT <= XT: checked by the launcher)
    const int L = L1 + K - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the wave's role in its quad, rotated from quad to quad: the four roles issue 184 / 184 / 152 / 116 MFMAs per tile
    // (edge positions have fewer taps, role 3 owns one dense output tile instead of two), and with three quads in lockstep
    // on a CU the SIMD that hosts role 0 of every quad would carry 552 of them -- rotated, the busiest carries 520
    const int quad = wave >> 2, q = ((wave & 3) + (p.rotate ? quad : 0)) & 3;
    const int g = lane >> 4, sq = lane & 15;
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem + p.total_floats);
    f4* xq = reinterpret_cast<f4*>(smem + p.total_floats + 64) + quad * (2 * XT * 64);   // this quad's two XT KiB buffers
    uint8_t* bytes_s = reinterpret_cast<uint8_t*>(smem + p.total_floats + 64 + QUADS * 2 * XT * 256) + quad * 256;   // dma: the first round's 16 x L bytes (L <= 16)
    fx_stamp(p.trace, 0);
    fx_stamp(p.trace, 7, (unsigned long long)fx_simd_id() + 1);
    [[maybe_unused]] unsigned tiles_done = 0;
    fx_lut_dma(lut_s, p.lut);                            // in flight together with the first member's weights

    int64_t u_lo, u_hi;
    int64_t Ncur = p.N, TGcur = p.TG;                    // SERVER: the current request's batch
    const uint8_t* ascii = SERVER ? p.min->bytes : p.ascii;
    float* outp = p.out;
    [[maybe_unused]] const int srv_m = SERVER ? (int)blockIdx.x / p.srv_tiles : 0;      // SERVER: workgroup = (member, tile slot)
    [[maybe_unused]] const int srv_slot = SERVER ? (int)blockIdx.x % p.srv_tiles : 0;
    if constexpr (SERVER) {
        u_lo = u_hi = 0;                                  // tiles come with the requests
    } else {
        fx_unit_range(p.TG, p.M, u_lo, u_hi);
        if (u_lo >= u_hi) return;
    }
    const int m_first = SERVER ? srv_m : (int)(u_lo / p.TG), m_last = SERVER ? srv_m : (int)((u_hi - 1) / p.TG);
    bool bad = false;
    int parity = 0;
    [[maybe_unused]] unsigned long long srv_last = 0, srv_start = 0, srv_seen = 0;
    // SERVER: request word, exit flag and the request's sequences (one read over PCIe) behind the quads' byte rows
    uint8_t* srv_area = reinterpret_cast<uint8_t*>(smem + p.total_floats + 64 + QUADS * 2 * XT * 256) + QUADS * 256;
    unsigned long long& srv_req = *reinterpret_cast<unsigned long long*>(srv_area);
    int& srv_exit = *reinterpret_cast<int*>(srv_area + 8);
    volatile int& srv_bad = *reinterpret_cast<volatile int*>(srv_area + 12);
    volatile int& srv_abandon = *reinterpret_cast<volatile int*>(srv_area + 16);        // a streamed request whose rows never came: not answered
    unsigned* srv_bytes = reinterpret_cast<unsigned*>(srv_area + 32) + quad * 256;      // this quad's tile: 16 x L bytes (<= 1 KiB)
    // dma: the very first round reads its bytes from LDS (a compiler-tracked byte load from global memory would be waited
    // for with vmcnt(0), i.e. together with the whole image); needs the 4-byte alignment of the dword copy
    const bool lds_bytes = !SERVER && p.dma && (reinterpret_cast<uintptr_t>(p.ascii) & 3) == 0;

    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();
        if (p.dma) {
            if (lds_bytes && m == m_first && q == 0) {
                const int64_t tg0 = u_lo - (int64_t)m * p.TG + quad;
                const int64_t rows = Ncur - tg0 * 16 < 16 ? Ncur - tg0 * 16 : 16;
                // (16 L bytes per tile: a multiple of 4; the last dword of a short tile may reach <= 3 bytes past the batch,
                //  inside the same aligned dword -- never into another page)
                if (u_lo + quad < u_hi && tg0 < p.TG && lane * 4 < rows * L)
                    fx_dma4(ascii + tg0 * 16 * L + lane * 4, __builtin_amdgcn_readfirstlane(fx_lds_addr(bytes_s)));
            }
            // conv part (+ conv1 rows, conv biases), then the head: both in flight, only the first is waited for here
            fx_dma_fill(smem, p.w[m], p.off_d1 / 4, QWAVES);
            const int head_loads = fx_dma_fill(smem + p.off_d1, p.w[m] + p.off_d1, (p.total_floats - p.off_d1) / 4, QWAVES);
            fx_wait_vm(head_loads);
        } else {
            fill_lds(reinterpret_cast<f4*>(smem), reinterpret_cast<const f4*>(p.w[m]), p.total_floats / 4);
            fx_wait_vm(0);
        }
        __syncthreads();
        if (m == m_first) fx_stamp(p.trace, 1);
        const f4* w_c2 = reinterpret_cast<const f4*>(smem + p.off_c2);
        const f4* w_c3 = reinterpret_cast<const f4*>(smem + p.off_c3);
        const float* cb = smem + p.off_cb;
        const float* w1p = smem + p.off_w1p;
        const f4* w_d1 = reinterpret_cast<const f4*>(smem + p.off_d1);
        const f4* w_d2 = reinterpret_cast<const f4*>(smem + p.off_d2);
        const float* db = smem + p.off_db;
        int64_t t_lo = SERVER ? 0 : (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        int64_t t_hi = SERVER ? 0 : (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        if constexpr (SERVER) {
            if (p.dma) fx_wait_vm(0);                                // the whole image before the first request
            __syncthreads();
            if (wave == 0) srv_start = srv_seen = wall_clock64();      // (every lane of wave 0 keeps the wait's clocks: fx_server_wait_line)
            if (tid == 0)
                __hip_atomic_store(const_cast<unsigned*>(&p.mout->alive[p.m_off + m][srv_slot]), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      for (;;) {                                                     // (SERVER: one iteration per request)
        [[maybe_unused]] int srv_rounds = 0;
        if constexpr (SERVER) {
            if (wave == 0) {
                int ex = 0;
                unsigned long long r = 0;
                if (srv_slot == 0) {
                    // the slot of tile 0 polls the whole request line: a tiny request's bytes arrive with the word (FxMailIn::tiny)
                    unsigned payload = 0;
                    r = fx_server_wait_line(p.min, srv_last, srv_seen, srv_start, p.idle_ticks, p.life_ticks, lane, &ex, &payload);
                    // (only the dwords that hold the request's N x L bytes: N <= 16, one tile)
                    if (!ex && (r & FX_SERVE_TINY) && lane >= 2 && lane < 14 && (lane - 2) * 4 < (int)(r & 0x3FFFull) * L) srv_bytes[lane - 2] = payload;
                } else if (lane == 0) {
                    r = fx_server_wait(p.min, srv_last, srv_seen, srv_start, p.idle_ticks, p.life_ticks, srv_slot < p.srv_fast, p.srv_sleep, &ex);
                }
                if (lane == 0) { srv_req = r; srv_exit = ex; srv_bad = 0; srv_abandon = 0; }
            }
            __syncthreads();
            if (srv_exit) break;
            Ncur = (int64_t)(srv_req & 0x3FFFull);
            TGcur = (Ncur + 15) >> 4;
            // this slot's tiles of the request: slot, slot + T, slot + 2 T, ... (T = slots per member) -- QUADS of them per round,
            // one per quad: a request of up to T tiles keeps one quad per workgroup busy (a lone quad is the fastest round),
            // a larger one fills the second and third quads before anybody needs a second round
            srv_rounds = srv_slot < TGcur ? (int)((TGcur - srv_slot + (int64_t)p.srv_tiles * QUADS - 1) / ((int64_t)p.srv_tiles * QUADS)) : 0;
            t_lo = srv_slot; t_hi = srv_slot + srv_rounds;
            bad = false;
            if (srv_rounds == 0) {                                   // a request with fewer tiles: nothing to answer from this slot
                if (wave == 0) { srv_last = srv_req; srv_seen = wall_clock64(); }
                __syncthreads();                                     // (everybody has read the request word)
                continue;
            }
        }
        const int rounds = SERVER ? srv_rounds : (int)((t_hi - t_lo + QUADS - 1) / QUADS);

        for (int rd = 0; rd < rounds; ++rd, parity ^= 1) {
            const int64_t tg = SERVER ? t_lo + ((int64_t)rd * QUADS + quad) * p.srv_tiles : t_lo + (int64_t)rd * QUADS + quad;
            const bool live = SERVER ? tg < TGcur : tg < t_hi;       // idle quads run along for the barriers
            const int64_t n = tg * 16 + sq;
            const uint8_t* row = ascii + ((live && n < Ncur) ? n : 0) * L;
            if constexpr (SERVER) {
                // this tile's bytes: one dword per thread (past the caches: the host wrote them through the BAR), then everybody
                // reads LDS.  (The previous round's readers of srv_bytes -- its phase A -- are several barriers behind.)
                const int64_t srv_rows = Ncur - tg * 16 < 16 ? Ncur - tg * 16 : 16;
                const int qt = tid & 255;
                if (live && (int64_t)qt * 4 < srv_rows * L && !(srv_req & FX_SERVE_TINY)) {          // (a tiny request's bytes came with the word)
                    // (a streamed request: the host is still packing; one that was given up is not waited for again)
                    if (!srv_abandon && !fx_server_rows_ready(p.min, srv_req, tg * 16 + srv_rows)) srv_abandon = 1;
                    srv_bytes[qt] = __hip_atomic_load(reinterpret_cast<const unsigned*>(ascii + tg * 16 * L) + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __syncthreads();
            }
            f4* X = xq + (parity ? XT * 64 : 0);
            f4* Y = xq + (parity ? 0 : XT * 64);
            asm volatile("" ::: "memory");                           // keep the LDS weight reads inside the round
            if (live && tiles_done == 0) fx_stamp(p.trace, 2);

            // ---- A: conv1 (valid) at this wave's positions: bias + the K kernel rows selected by the codes, relu
            if (live) {
                for (int pos = q; pos < L1; pos += 4) {
                    f4 o1[FT][1];
                    int c[K];
                    auto codes = [&](auto rp) {
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            c[j] = lut_s[rp[pos + j]];
                            if (c[j] == 0xFF) { bad = true; c[j] = 0; }
                        }
                    };
                    if constexpr (SERVER) codes((fx_lds_u8p)(reinterpret_cast<uint8_t*>(srv_bytes) + ((live && n < Ncur) ? sq : 0) * L));
                    else if (lds_bytes && m == m_first && rd == 0) codes((fx_lds_u8p)(bytes_s + (n < Ncur ? sq : 0) * L));
                    else codes(row);
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
            if constexpr (SERVER) { if (bad) srv_bad = 1; }          // (read by wave 0 when it answers, several barriers later)
            __syncthreads();
            if (live) FX_PHASE_STAMP(8);

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
            if (p.dma && rd == 0) fx_wait_vm(0);                     // this wave's share of the head's weights has landed
            __syncthreads();
            if (live) FX_PHASE_STAMP(9);

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
                                if (mi == HT - 1 && r >= p.rlh) break;
                                acc = mfma16(a[r], h1[mi][r], acc);
                            }
                        }
                        X[mo * 64 + lane] = relu4(acc);
                    }
                }
            }
            __syncthreads();
            if (live) FX_PHASE_STAMP(10);

            // ---- F: Dense(1) on wave 0 of the quad
            if (live && q == 0) {
                f4 h2[HT][1];
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) h2[mi][0] = X[mi * 64 + lane];
                float y[1];
                final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2, y, g);
                if constexpr (SERVER) {
                    // (score, tag) in one 8-byte SYSTEM-scope store to host memory: written through every cache level by itself
                    // (round 3 added a system fence per tile -- ~0.5 us each, serialised per XCD; off by default now, serve_fence)
                    const unsigned tag = (unsigned)(srv_req >> 16) | (srv_bad ? 0x80000000u : 0u);
                    if (g == 0 && n < Ncur && !srv_abandon)
                        __hip_atomic_store(const_cast<unsigned long long*>(&p.mout->ans[p.m_off + m][n]),
                                           ((unsigned long long)tag << 32) | __float_as_uint(fx_nan_to_num(y[0])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (p.srv_fence) __threadfence_system();
                } else {
                    if (g == 0 && n < Ncur) outp[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[0]);
                }
#if defined(FX_AB)   // (the in-kernel mean of explorer-size launches: no faster than the 3 us mean launch, A/B build only)
                if (!SERVER && p.mean_out) {
                    // np.mean over the members (ensemble.py:24) without a second launch: publish this member's 16 scores
                    // device-wide, take the tile's ticket; the M-th arrival reads all members' scores back (past its own
                    // L2: other members' workgroups may sit on other XCDs) and averages in NumPy's order.  Once per tile
                    // and member, on launches of a few tiles -- the cross-XCD round trips that made this ruinous per work
                    // unit of a 1e5-sequence launch cost a fraction of a microsecond here and save
                    // the ~4 us mean launch of every explorer-size call.
                    __threadfence();
                    unsigned t = 0;
                    if (lane == 0) t = __hip_atomic_fetch_add(&p.tickets[tg], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
                    if (t == (unsigned)p.M - 1u) {
                        __threadfence();
                        if (g == 0 && n < Ncur) {
                            float x[16];
#pragma unroll
                            for (int mm = 0; mm < 16; ++mm)
                                x[mm] = mm < p.M ? __hip_atomic_load(p.out + n * p.out_sn + mm * p.out_sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                            p.mean_out[n] = np_mean_row16(x, p.M);
                        }
                        if (lane == 0) __hip_atomic_store(&p.tickets[tg], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#endif
            }
            if (live) FX_TILE_DONE();
        }
        if constexpr (SERVER) {
            if (wave == 0) { srv_last = srv_req; srv_seen = wall_clock64(); }
            __syncthreads();
        } else {
            break;
        }
      }
        if constexpr (SERVER) {
            if (tid == 0) __hip_atomic_store(const_cast<unsigned*>(&p.mout->alive[p.m_off + m][srv_slot]), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    fx_stamp(p.trace, 6);
    if (!SERVER && bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

}  // namespace

namespace {

template <int QUADS, int XT, int L1C, int K = 5, int HT = 7>
int launch_quad(fx_engine* e, QuadArgs a, int64_t U, int max_rounds) {
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    if (blocks > U) blocks = U;
    if (e->cnn_quad < 2 && U > (int64_t)max_rounds * QUADS * blocks) return FX_EUNSUPPORTED;   // long launches: one wave per tile
    const size_t lds = (size_t)a.total_floats * 4 + 256 + (size_t)QUADS * 2 * XT * 1024 + (size_t)QUADS * 256;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    auto kern = k_score_cnn_quad<HT, QUADS, XT, L1C, K>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(QUADS * 256), lds, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

}  // namespace

// FX_EUNSUPPORTED when the quad form does not apply (the caller carries on with the one-wave-per-tile kernels).
int fx_launch_score_cnn_quad(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                             float* d_out_NM, int Mtot, int m_off) {
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    const int L1 = s.L - s.K + 1;
    // hidden layer: 7 tiles (97-112 units) for every kernel size; 1 / 2 / 4 tiles (<= 64 units) for kernel size 5
    const bool ht_ok = lay.HT == 7 || (s.K == 5 && (lay.HT == 1 || lay.HT == 2 || lay.HT == 4));
    if (!e->cnn_quad || s.A != 4 || (s.K != 5 && s.K != 3 && s.K != 7) || L1 < 1 || L1 > 12 || lay.FT != 2 || !ht_ok ||
        e->cnn_conv1_mfma || e->cnn_variant || M > FX_MAX_M)
        return FX_EUNSUPPORTED;
    const int64_t TG = (N + 15) / 16, U = (int64_t)M * TG;
    QuadArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    if (int rc = fx_trace_buffer(e, &a.trace)) return rc;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.N = N; a.TG = TG; a.M = M; a.m_off = m_off;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.L = s.L; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.rotate = (int)e->quad_rotate;
    a.dma = e->dma_fill && lay.off_d1 % 4 == 0 && lay.total_floats % 4 == 0;
#if defined(FX_AB)
    const bool fuse = e->fuse_mean_out && e->planar_stride && m_off == 0 && M == Mtot && M <= 16 && TG <= 4096;
#else
    const bool fuse = false;
#endif
    if (fuse) {
        void* tk = nullptr;
        if (int rc = fx_zero_pool(e, sizeof(unsigned) * (size_t)TG, &tk)) return rc;
        a.mean_out = e->fuse_mean_out;
        a.tickets = (unsigned*)tk;
    }
    const int rc_q = [&]() -> int {
    a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3; a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p;
    a.off_d1 = (int)lay.off_d1; a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db; a.total_floats = (int)lay.total_floats;
    // up to 4 conv positions (seq_len <= 8): three quads per workgroup, one round; up to 12 (seq_len <= 16): one quad
    // with 2 x 24 KiB of exchange buffers, up to two rounds (a round is ~9 us against ~30 us for a lone wave's tile)
    // (kernel sizes 3 and 7, the other two fused instantiations of the one-wave kernel: one quad, any position count)
    if (s.K == 3) return launch_quad<1, 24, 0, 3>(e, a, U, 2);
    if (s.K == 7) return launch_quad<1, 20, 0, 7>(e, a, U, 2);     // (seq_len <= 16: at most 10 positions; its image is 8 KiB larger)
    switch (lay.HT) {
        case 1: return L1 <= 4 ? launch_quad<3, 8, 0, 5, 1>(e, a, U, 1) : launch_quad<1, 24, 0, 5, 1>(e, a, U, 2);
        case 2: return L1 <= 4 ? launch_quad<3, 8, 0, 5, 2>(e, a, U, 1) : launch_quad<1, 24, 0, 5, 2>(e, a, U, 2);
        case 4: return L1 <= 4 ? launch_quad<3, 8, 0, 5, 4>(e, a, U, 1) : launch_quad<1, 24, 0, 5, 4>(e, a, U, 2);
        default: break;
    }
    if (L1 == 4) return launch_quad<3, 8, 4>(e, a, U, 1);
    if (L1 < 4) return launch_quad<3, 8, 0>(e, a, U, 1);
    return launch_quad<1, 24, 0>(e, a, U, 2);
    }();
    if (rc_q == FX_OK && fuse) e->fused_mean_done = true;
    return rc_q;
}


// ---- the resident form ----------------------------------------------------------------------------------------------
namespace {

template <int QUADS, int XT, int L1C>
int launch_server(fx_engine* e, QuadArgs a, int M, hipStream_t stream) {
    const size_t lds = (size_t)a.total_floats * 4 + 256 + (size_t)QUADS * 2 * XT * 1024 + (size_t)QUADS * 256 + 32 + (size_t)QUADS * 1024;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    auto kern = k_score_cnn_quad<7, QUADS, XT, L1C, 5, true>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(M * a.srv_tiles)), dim3(QUADS * 256), lds, stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

}  // namespace

// Canonical shapes only (what the explorers' surrogates are built with: kernel size 5, 32 filters, 97-112 hidden units,
// 4-letter alphabet, seq_len <= 16).  One workgroup (one quad) per member and tile slot: a request of N sequences is
// answered by M x min(ceil(N / 16), tiles) of them, each on its own CU, slot s walking the tiles s, s + tiles, ... (round 4:
// a wide generation holds most of the chip and serves up to 4096 sequences per request; round 3's held a third of it).
int fx_launch_score_cnn_quad_server(fx_engine* e, fx_model* const* models, int M, int m_off, int tiles, hipStream_t stream,
                                    FxMailIn* d_in, FxMailOut* d_out, unsigned long long idle_ticks, unsigned long long life_ticks,
                                    int want_quads, int* quads_out) {
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    const int L1 = s.L - s.K + 1;
    if (s.kind != FX_CNN || s.A != 4 || s.K != 5 || L1 < 1 || L1 > 12 || lay.FT != 2 || lay.HT != 7 || M > FX_MAX_M || M < 1 ||
        e->cnn_conv1_mfma || e->cnn_variant)
        return FX_EUNSUPPORTED;
    for (int m = 1; m < M; ++m) {
        const FxShape& t = models[m]->shape;
        if (t.kind != s.kind || t.L != s.L || t.A != s.A || t.F != s.F || t.H != s.H || t.K != s.K) return FX_EUNSUPPORTED;
    }
    QuadArgs a{};
    a.lut = e->d_lut; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.M = M; a.L = s.L; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.dma = e->dma_fill && lay.off_d1 % 4 == 0 && lay.total_floats % 4 == 0;
    a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3; a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p;
    a.off_d1 = (int)lay.off_d1; a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db; a.total_floats = (int)lay.total_floats;
    a.min = d_in; a.mout = d_out; a.idle_ticks = idle_ticks; a.life_ticks = life_ticks;
    a.srv_tiles = tiles; a.m_off = m_off; a.srv_fast = e->server.fast; a.srv_sleep = (int)e->serve_poll_sleep; a.srv_fence = (int)e->serve_fence;
    // want_quads = 3 (a wide generation, seq_len <= 8): up to three tiles per workgroup side by side, like the launched form.  The
    // second and third quads only get tiles when a request has more tiles than the generation has workgroups: a round with one
    // busy quad takes ~6 us, with two ~8.5 us (four waves per tile, ~200 fp32 MFMAs each: the quads share the CU's four MFMA
    // pipes), against 12 us for two rounds -- 44.8 -> 41.2 us for a 4096-sequence call of the 3 x CNN ensemble, 27.5 -> 26.6 us for
    // 2001 (profiles/r4_server_quads_ab.log).  Filling the quads of ONE workgroup first was measured too: +2.4 us for every
    // 20-sequence call (same log, first table).
    // Once requests are STREAMED (tiles start as their rows arrive, staggered anyway) the extra quads lose: 23.3 vs 25.4 us for
    // 2001 sequences, 25.8 vs 30.6 us for 4096 of a single CNN.  A/B build only.
    *quads_out = 1;
#if defined(FX_AB)
    if (want_quads >= 3 && L1 <= 4) {
        a.rotate = (int)e->quad_rotate;
        const int rc = L1 == 4 ? launch_server<3, 8, 4>(e, a, M, stream) : launch_server<3, 8, 0>(e, a, M, stream);
        if (rc != FX_EUNSUPPORTED) { *quads_out = 3; return rc; }
    }
#else
    (void)want_quads;
#endif
    if (L1 == 4) return launch_server<1, 8, 4>(e, a, M, stream);
    if (L1 < 4) return launch_server<1, 8, 0>(e, a, M, stream);
    return launch_server<1, 24, 0>(e, a, M, stream);
}
