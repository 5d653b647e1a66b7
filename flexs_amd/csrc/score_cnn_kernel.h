// K1 score_cnn: string -> one-hot -> Conv1D x3 -> GlobalMaxPool -> Dense x3, fused, on f32 MFMA.
//
// Replaces, per (sequence, ensemble member):  sequence_utils.py:32-47 (encode),
// keras_model.py:69-79 (tensor + predict + nan_to_num), cnn.py:23-54 (layers).
//
// One wave owns NT tiles of 16 sequences and streams over sequence positions,
// keeping sliding windows of the conv1 / conv2 outputs in registers (see
// mfma_common.h for the transposed-MFMA formulation).  Weights of the member
// being scored sit in LDS in fragment order (pack.cpp); work units are
// (member, tile-group) pairs, flattened member-major and split evenly over the
// grid, so a block reloads LDS at most once per member boundary it straddles.
//
// Algorithmic work per sequence per member: L + 4 bytes of HBM
// traffic, 2 * MACs FLOP with MACs = L1*K*A*F + L1*K*F*F + L1*(A-1)*F*F + F*H + H*H + H.
// The kernel is f32-MFMA-bound (157.3 TFLOP/s peak), not HBM-bound.
// The fused CNN scoring kernel template (shared by score_cnn_mfma.hip, which instantiates the fused forms, and
// score_cnn_split.hip, which instantiates the conv-only form for non-canonical shapes).
#pragma once
#include "fx_common.h"
#include "mfma_common.h"
#include "score_cnn_quad_round.h"
#include "np_sum.h"

namespace {

struct CnnArgs {
    const uint8_t* ascii;       // N x L
    const uint8_t* lut;         // 256
    const float* w[FX_MAX_M];   // packed weights per member
    float* out;                 // N x Mtot
    unsigned* err;
    unsigned long long* trace;  // in-kernel timeline (null = off), see fx_stamp
    int wave_prio;              // 1 = fx_stagger_priority
    int stage_fill;             // 1 = small launches load the conv part first, idle waves bring the head's weights
    int quad_tail;              // QT: 1 = the last (tiles mod 4) tiles of a workgroup are walked by wave quads (fx_cnn_quad_round)
    int64_t N;
    int64_t TG;                 // tile groups per member = ceil(N / (16*NT))
    int M, Mtot, m_off;
    int64_t out_sn, out_sm;     // score of (sequence n, member column c) lives at out[n * out_sn + c * out_sm]
    f4* pool_out;               // HEAD = false: pooled conv features, [(member * TG + tile) * FT + t][64 lanes]
    int L;
    int rlh;                    // hidden tail: real k-steps of the last real hidden tile
    // SEG with seg_sb > 1: a tile's positions are cut over seg_sb workgroups (grid = units x seg_sb); their maxima meet in
    // seg_pool ([unit][FT][64 lanes][4], float bits; fx_zero_pool: all zeros between launches, the last workgroup to arrive
    // (seg_cnt ticket) reads the maxima, resets the entries and runs the head
    int seg_sb;
    unsigned* seg_pool;
    unsigned* seg_cnt;
    // packed-layout offsets (floats)
    int off_first, off_c2, off_c3, off_cb, off_w1p, conv_floats, off_d1, off_d2, off_db, total_floats;
    int stw_stride;             // STG: bytes of LDS scratch per wave (16 L rounded up to 16)
    int stw_ahead;              // STG: 1 = two scratches per wave, the next tile's bytes are asked for a tile ahead (fx_stage_tile_dma)
    FxRowsReady ready;          // launched-first host call: the rows arrive while the kernel runs (words == nullptr: they are all there)
#if defined(FX_AB)
    // fused ensemble mean (A/B build; fm_mean == nullptr: off): `out` is M member-major planes `out_sm` floats apart, all members in this launch
    float* fm_mean;             // np.mean over the members, N floats
    unsigned* fm_cnt;           // one ticket counter per tile, all zeros between launches (fx_zero_pool)
#endif
};

#if defined(FX_AB)
// Fused ensemble mean (ensemble.py:54-59 with the default np.mean; replaces the k_ensemble_mean_planar launch behind a batch launch).
// MEASURED AND LOST (round 6, profiles/r6_fused_mean_ab.log: the headline step 183.5 -> 185.1 us; a wave waits ~2 us per tile for its
// written-through scores and its ticket, which costs the launch more than the 3.6 us mean kernel behind it): A/B build only.
// The wave that finishes member m of a tile writes its 16 scores THROUGH to memory (sc1 stores), waits for them, and takes a ticket
// from the tile's counter (relaxed, device scope); the wave that draws the LAST ticket reads the other members' scores with device-scope
// loads -- written through and waited for before their tickets were drawn -- and stores the NumPy-order mean (np_sum_row's order for fewer than eight
// members, fx_np_div: the mean kernel's arithmetic, so the same bits), then puts the counter back to zero for the next launch.  No fences: a release /
// acquire pair at device scope writes back and invalidates the XCD's whole L2 (round 1, run 42: 0.195 -> 0.499 ms per step).
__device__ __forceinline__ void fx_fused_mean_tile(float* planes, int64_t stride, int M, int m, int64_t tile, int64_t n, int64_t N,
                                                   float y, bool holder, float* mean, unsigned* cnt, int lane) {
    unsigned* pl = reinterpret_cast<unsigned*>(planes);
    if (holder && n < N) __hip_atomic_store(&pl[(int64_t)m * stride + n], __float_as_uint(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the wave's scores are in memory before its ticket is drawn
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (unsigned)M - 1u) return;
    if (lane == 0) __hip_atomic_store(&cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (holder && n < N) {
        // M < 8: NumPy sums a row of fewer than eight floats front to back from 0 (np_sum_row; seven values in flight, no array)
        float x[7];
#pragma unroll
        for (int k = 0; k < 7; ++k)
            x[k] = k < M ? (k == m ? y : __uint_as_float(__hip_atomic_load(&pl[(int64_t)k * stride + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) : 0.f;
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k < M) r += x[k];
        mean[n] = fx_np_div(r, (float)M);
    }
}
#define FX_FM_ON(p) ((p).fm_mean != nullptr)
#else
#define FX_FM_ON(p) false
#endif

// L1S > 0: number of conv positions known at compile time (L1S = seq_len - K + 1): the position loop is fully
// unrolled, so the sliding windows become register renames instead of v_mov chains and the padding tests fold.
// SEG: small batches (fewer tiles than CUs) -- the WAVES waves of a workgroup share ONE tile and split its conv
// positions: wave q streams segment q plus a halo of PL3 + PL2 positions before and PR2 + PR3 after it, pools
// only its own positions (every conv3 output sees the same MFMA sequence as in the whole-sequence walk, so the
// result is bit-identical), the segment maxima meet in LDS and wave 0 runs the dense head.
// HEAD = false: the conv part only -- the global max-pooled features of every tile go to `pool_out` in accumulator
// (= B-operand) layout and a separate head kernel (score_cnn_split.hip) finishes the sequence.  Used for the CNN
// shapes whose (kernel size, hidden width) pair has no fused instantiation.
// QT (round 6; the unrolled seq_len = 8 form in 16-wave workgroups): 18.31 tiles per SIMD on the 3 x 1e5 headline launch mean that a SIMD
// with a 19th tile sets the launch (in-kernel timeline profiles/r6_trace_probe.json: workgroups of 72 tiles leave at 175 us, those of 73 / 74
// at 192 us).  The (tiles mod 4) last tiles of a workgroup are left out of the per-SIMD shares and walked by wave QUADS -- one wave per SIMD,
// a quarter of the tile's MFMAs on each pipe -- in one round after the main loop (fx_cnn_quad_round: same bits).
// Measured with it and NOT kept: a staged start (image by direct global -> LDS copies in two parts, every wave's first row asked for ahead of
// the copies, first tiles start when the 35 KiB conv part has landed, the head part awaited through an LDS count) -- 1.5 % SLOWER on the
// headline launch and on a one-member launch (profiles/r6_dma_start_ab.log; tools/archive/runs/r6_dma_start_ab.py).
template <int A, int K, int FT, int HT, int NT, bool DENSE_LDS, int WAVES, bool G1, int L1S = 0, bool PRIO = false,
          bool SEG = false, bool HEAD = true, bool STG = false, bool QT = false>
__global__ void __launch_bounds__(WAVES * 64) k_score_cnn_mfma(CnnArgs p) {
    static_assert(!QT || (A == 4 && FT == 2 && HT == 7 && NT == 1 && DENSE_LDS && WAVES == 16 && G1 && L1S > 0 && L1S <= 4 && !SEG && HEAD && !STG),
                  "QT is a form of the canonical unrolled kernel (exchange buffers of 8 KiB: four conv positions)");
    constexpr int QXT = 8;                                  // QT: 16-channel tiles per exchange buffer (>= FT * L1S, >= HT + 1)
    static_assert(!STG || (!SEG && NT == 1), "STG copies ONE whole tile's bytes into the wave's LDS scratch");
    static_assert(HEAD || (NT == 1 && !DENSE_LDS), "the conv-only form is one tile per wave, conv weights in LDS");
    static_assert(!SEG || (NT == 1 && L1S == 0 && K * (A - 1) <= 16), "SEG is the ring-window, one-tile form");
    constexpr int K3 = A - 1;
    constexpr int PL2 = (K - 1) / 2, PR2 = K - 1 - PL2;
    constexpr int PL3 = (K3 - 1) / 2, PR3 = K3 - 1 - PL3;
    constexpr int S1 = (K * A + 3) / 4;                 // conv1 k-steps
    static_assert(G1 || A % 4 == 0, "MFMA form of the one-hot conv: k-steps must not straddle a tap");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L, L1 = L1S > 0 ? L1S : L - K + 1;
    const int lds_floats = DENSE_LDS ? p.total_floats : p.conv_floats;
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem + lds_floats);
    int* next_tile = reinterpret_cast<int*>(smem + lds_floats + 64);   // 4 work counters (one per SIMD), after the 256-byte LUT
    int* simd_waves = next_tile + 4;                                   // 4 wave counts (workgroup's waves per SIMD)
    int* stage_ctl = simd_waves + 4;                                   // staged fill: [0] next chunk, [1] chunks in LDS
    // STG (the sequences lie in HOST memory, read over PCIe -- the zero-copy host calls): the 16 x L bytes of a tile are brought into
    // this wave's LDS scratch with ONE 16-byte load per lane and the position loop picks its bytes from there.  Reads of host memory
    // are served by no cache: the byte load per position and lane group of the form below was a PCIe round trip each (~2 us), which
    // four waves per SIMD hide in a long launch but not at 1-2 tiles per wave (profiles/r5_e2e_breakdown.log: 98 us against 62 us for
    // the 1e5-sequence launch with the bytes in HBM).
    [[maybe_unused]] uint8_t* stw = reinterpret_cast<uint8_t*>(smem + lds_floats + 64 + 12) + (tid >> 6) * p.stw_stride * (p.stw_ahead ? 2 : 1);
    [[maybe_unused]] uint8_t* stw_alt = stw + p.stw_stride;

    fx_stamp(p.trace, 0);
    if (p.wave_prio) fx_stagger_priority();
    const int simd = fx_simd_id();
    fx_stamp(p.trace, 7, (unsigned long long)simd + 1);
    for (int i = tid; i < 64; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];
    if (tid < 4) simd_waves[tid] = 0;

    int64_t u_lo, u_hi;
    const int seg_sb = (SEG && HEAD && p.seg_sb > 1) ? p.seg_sb : 1;   // workgroups per tile (multi-workgroup segment form)
    const int seg_w = SEG ? (int)(blockIdx.x % (unsigned)seg_sb) : 0;
    if (SEG && seg_sb > 1) { u_lo = blockIdx.x / (unsigned)seg_sb; u_hi = u_lo + 1; }
    else fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    bool bad = false;
    [[maybe_unused]] unsigned tiles_done = 0;
    FxSimdShare share{0, 1, 1};

    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();                                 // previous member's readers are done
        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        // Staged fill (small launches: at most half as many tiles as waves).  Only the conv part (~35 of ~103 KiB) is loaded before
        // the first tile starts; the waves that get no tile bring the dense head's weights meanwhile, chunk by chunk
        // from an LDS counter, and the tile waves wait for the chunk count before their head -- an LDS flag instead of
        // a barrier, because the waves are in different places.  The LDS fill is a fifth of a small launch.
        constexpr bool CAN_STAGE = DENSE_LDS && HEAD && !SEG && WAVES <= 8;   // (the 16-wave forms run long launches and have no registers to spare)
        const bool staged = CAN_STAGE && p.stage_fill && 2 * (t_hi - t_lo) <= WAVES;   // at least half the waves are idle
        const int dense_f4 = (lds_floats - p.conv_floats) / 4, n_chunks = (dense_f4 + 511) / 512;
        if (tid < 4) next_tile[tid] = 0;
        if (tid < 2) stage_ctl[tid] = 0;
        if (m == m_first) fx_count_simd_wave(simd_waves, simd);      // (zeroed before the barrier above)
        {
            const f4* src = reinterpret_cast<const f4*>(p.w[m]);
            f4* dst = reinterpret_cast<f4*>(smem);
            fill_lds(dst, src, (staged ? p.conv_floats : lds_floats) / 4);   // (14 loads in flight per thread instead of 8: 3.4 us instead of 3.0, r2 trace)
        }
        __syncthreads();
        if (m == m_first) share = fx_simd_share(simd_waves, simd);
        if (m == m_first) fx_stamp(p.trace, 1);
        const f4* w_first = reinterpret_cast<const f4*>(smem + p.off_first);
        const f4* w_c2 = reinterpret_cast<const f4*>(smem + p.off_c2);
        const f4* w_c3 = reinterpret_cast<const f4*>(smem + p.off_c3);
        const float* cb = smem + p.off_cb;
        const float* w1p = smem + p.off_w1p;
        const float* dbase = DENSE_LDS ? smem : p.w[m];
        const f4* w_d1 = reinterpret_cast<const f4*>(dbase + p.off_d1);
        const f4* w_d2 = reinterpret_cast<const f4*>(dbase + p.off_d2);
        const float* db = dbase + p.off_db;   // (not const-qualified pointers: laundered per tile below)
        bool got_tile = false, head_ready = !staged;

        // the workgroup's tiles in shares per SIMD (proportional to the waves it hosts); the waves of a SIMD pull from
        // their share's counter
        // QT: the tiles that do not divide by four wait for the quad round behind the main loop (not when the rows still arrive)
        int nq = 0;
        if constexpr (QT) {
            if (p.quad_tail && t_hi - t_lo >= 8 && !p.ready.words) nq = (int)((t_hi - t_lo) & 3);
        }
        const int64_t t_main = t_hi - nq;
        const int64_t s_lo = t_lo + (t_main - t_lo) * share.before / share.total;
        const int64_t s_hi = t_lo + (t_main - t_lo) * (share.before + share.mine) / share.total;
        // launched-first call: the host packs the tiles of stage 0 first, then stage 1, ... (FxRowsReady): the share is walked from
        // its first tile of stage 0 on (and around), so the stages it asks for come in the order they are packed
        const bool rows_arrive = !SEG && NT == 1 && p.ready.words != nullptr;
        int rows_rot = 0, rows_known = 0;
        if (rows_arrive) {
            const int64_t t0 = (s_lo + p.ready.Q - 1) / p.ready.Q * p.ready.Q;
            if (t0 < s_hi) rows_rot = (int)(t0 - s_lo);
        }

        // the next tile of this wave's SIMD share (-1: none left)
        auto pull_tile = [&]() -> int64_t {
            int pulled = 0;
            if (lane == 0) pulled = atomicAdd(&next_tile[simd], 1);
            pulled = __builtin_amdgcn_readfirstlane(pulled);
            if (s_lo + pulled >= s_hi) return -1;
            if (!rows_arrive) return s_lo + pulled;
            const int len = (int)(s_hi - s_lo);
            int at = pulled + rows_rot;
            if (at >= len) at -= len;
            return s_lo + at;
        };
        [[maybe_unused]] int64_t pf_next = -2;              // STG, stw_ahead: the tile claimed ahead (-2: none claimed, -1: the share is exhausted)
        [[maybe_unused]] bool pf_issued = false;            // ... and its bytes are on their way into stw_alt
        for (int64_t seg_tile = t_lo;; ++seg_tile) {
            int64_t tg = seg_tile;                               // SEG: every wave of the workgroup walks the same tiles
            if (!SEG) {
                tg = (STG && pf_next != -2) ? pf_next : pull_tile();
                if (tg < 0) break;
                if (rows_arrive) fx_rows_wait(p.ready, (int)(tg % p.ready.Q), rows_known, p.err);
            }
            if (tg >= t_hi) break;
            got_tile = true;
            if (tiles_done == 0) fx_stamp(p.trace, 2);
            // ---- this lane's sequences
            int64_t n[NT];
            const uint8_t* row[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                n[nt] = (tg * NT + nt) * 16 + sq;
                row[nt] = p.ascii + (n[nt] < p.N ? n[nt] : 0) * L;   // out-of-range lanes recompute seq 0
                if (rows_arrive) row[nt] = p.ascii + tg * p.ready.pitch + (n[nt] < p.N ? sq : 0) * L;   // (tile-pitched staging; a ragged last tile's spare lanes take its row 0)
            }
            [[maybe_unused]] fx_lds_u8p srow = nullptr;
            if constexpr (STG) {
                const int64_t tile_rows = p.N - tg * 16 < 16 ? p.N - tg * 16 : 16;
                const int64_t tile_pitch = rows_arrive ? (int64_t)p.ready.pitch : (int64_t)16 * L;
                if (pf_issued) {
                    // asked for a tile ago: the bytes are in the other scratch once the loads have landed
                    uint8_t* t = stw; stw = stw_alt; stw_alt = t;
                    fx_wait_vm(0);
                } else fx_stage_tile(p.ascii + tg * tile_pitch, (int)tile_rows * L, stw, lane);
                pf_next = -2; pf_issued = false;
                if (p.stw_ahead) {
                    // claim the next tile now; if it is whole and its rows are known to be there, ask for its bytes: the PCIe round
                    // trip then lies beside this tile's work instead of in front of the next one's
                    pf_next = pull_tile();
                    if (pf_next >= 0 && (pf_next + 1) * 16 <= p.N && (!rows_arrive || (int)(pf_next % p.ready.Q) < rows_known)) {
                        fx_stage_tile_dma(p.ascii + pf_next * tile_pitch, 16 * L, stw_alt, lane);
                        pf_issued = true;
                    }
                }
                srow = (fx_lds_u8p)(stw + (n[0] < p.N ? sq : 0) * L);
            }
            // a byte of this lane's sequence.  ROW8 (seq_len = 8, rows in device memory; round 6): the whole row is ONE 8-byte load per tile
            // (any alignment: gfx950 compute runs with unaligned access enabled) and the positions are shifts, instead of eight byte loads:
            // -1 % on the headline launch
            constexpr bool ROW8 = !STG && L1S == 4 && K == 5;
            [[maybe_unused]] unsigned long long rowq[NT];
            if constexpr (ROW8) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) __builtin_memcpy(&rowq[nt], row[nt], 8);
            }
            auto seq_byte = [&](int nt, int at) -> int {
                if constexpr (STG) return (int)srow[at];
                else if constexpr (ROW8) return (int)((rowq[nt] >> (8 * at)) & 0xFFull);
                else return (int)row[nt][at];
            };
            // Sliding windows.  RING: positions live in slot (position mod window), and the position loop is
            // unrolled by a multiple of both window lengths, so every slot index is a compile-time constant and
            // nothing is ever moved.  Otherwise (19-tap protein window, A/B baseline only) slots are shifted.
            constexpr bool RING = (L1S > 0) || (K * K3 <= 16);
            constexpr int UN = L1S > 0 ? (L1S + PR2 + PR3) : (RING ? K * K3 : 1);   // trips per unrolled block
            // code window: alphabet index at positions s .. s+K-1
            int cw[K][NT];
#pragma unroll
            for (int j = 0; j < K - 1; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    int c = lut_s[seq_byte(nt, j)];
                    if (c == 0xFF) { bad = true; c = 0; }
                    cw[RING ? j : j + 1][nt] = c;         // (shifting form: moved down at the top of step 0)
                }
            // Look-ahead queue of raw sequence bytes (generic-length form): the byte a step consumes was requested PF
            // steps earlier, so its L2 / HBM latency is off the step's critical path.  A lone wave (small batches:
            // one tile per SIMD) otherwise pays one global round trip per position (profiles/archive/r2_trace_probe: 8.4 us
            // for the conv part of an L = 8 tile whose MFMAs take 5.5 us).  The unrolled forms (L1S > 0) run with
            // four waves per SIMD, which hide it, and have no registers to spare.
            constexpr bool AHEAD = (L1S == 0) && (WAVES <= 8);     // (16-wave forms: four waves per SIMD hide it, no registers to spare)
            constexpr int PF = !AHEAD ? 1 : (!RING ? 2 : (UN % 3 == 0 ? 3 : (UN % 2 == 0 ? 2 : 1)));
            static_assert(!AHEAD || !RING || UN % PF == 0, "ring slots of the byte queue must be compile-time constants");
            int rq[PF][NT];
            f4 win1[K][FT][NT], win2[K3][FT][NT], gmax[FT][NT];
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int t = 0; t < FT; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) win1[j][t][nt] = splat4(0.f);
#pragma unroll
            for (int j = 0; j < K3; ++j)
#pragma unroll
                for (int t = 0; t < FT; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) win2[j][t][nt] = splat4(0.f);
#pragma unroll
            for (int t = 0; t < FT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) gmax[t][nt] = splat4(0.f);   // relu output >= 0

            const int steps = L1 + PR2 + PR3;
            // SEG: positions this wave pools, and the steps it has to run for them
            int seg_lo = 0, seg_hi = L1, s_begin = 0, s_stop = steps;
            if (SEG) {
                const int q = seg_w * WAVES + (tid >> 6), S = seg_sb * WAVES;
                seg_lo = (int)((int64_t)L1 * q / S);
                seg_hi = (int)((int64_t)L1 * (q + 1) / S);
                s_begin = seg_lo - PL3 - PL2 > 0 ? seg_lo - PL3 - PL2 : 0;
                s_stop = seg_hi + PR2 + PR3 < steps ? seg_hi + PR2 + PR3 : steps;
                if (seg_lo >= seg_hi) s_stop = 0;                 // more waves than positions: nothing to do
            }
            const int s_first = SEG ? (s_begin / UN) * UN : 0;     // ring slots assume block starts at multiples of UN
            if (SEG && s_first > 0) {
                // the code window is positional: refill it for the block this wave starts in
#pragma unroll
                for (int j = 0; j < K - 1; ++j) {
                    int c = lut_s[row[0][s_first + j]];
                    if (c == 0xFF) { bad = true; c = 0; }
                    cw[j][0] = c;
                }
            }
            if (AHEAD) {
#pragma unroll
                for (int q = 0; q < PF; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        rq[q][nt] = (s_first + K - 1 + q < L) ? seq_byte(nt, s_first + K - 1 + q) : 0;
            }
            for (int s0 = s_first; s0 < s_stop; s0 += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int s = s0 + u;
                const bool act = !SEG || s >= s_begin;             // SEG: steps before the halo only advance the code window
                if (L1S > 0 || s < s_stop) {
                // weights in LDS are loop-invariant: without this barrier LICM hoists every
                // ds_read out of the position loop and spills hundreds of VGPRs
                // (fencing only every 2nd / 4th position of the unrolled kernels measured no gain: profiles/archive/r1_run18)
                asm volatile("" ::: "memory");
                if (!RING) {
                    // ---- slide the windows
#pragma unroll
                    for (int j = 0; j < K - 1; ++j)
#pragma unroll
                        for (int t = 0; t < FT; ++t)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) win1[j][t][nt] = win1[j + 1][t][nt];
#pragma unroll
                    for (int j = 0; j < K3 - 1; ++j)
#pragma unroll
                        for (int t = 0; t < FT; ++t)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) win2[j][t][nt] = win2[j + 1][t][nt];
#pragma unroll
                    for (int j = 0; j < K - 1; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) cw[j][nt] = cw[j + 1][nt];
                }
                // slot of: code[s + j]; out1[s] (newest); out1[s - (K-1) + j]; out2[t2] (newest); out2[t2 - (K3-1) + j]
                // (s = u mod K and mod K3 because s0 is a multiple of both; all constants once unrolled)
#define FX_CW(j) (RING ? (u + (j)) % K : (j))
#define FX_W1NEW (RING ? u % K : K - 1)
#define FX_W1(j) (RING ? (u + 1 + (j)) % K : (j))
#define FX_W2NEW (RING ? (u + K3 * K - PR2) % K3 : K3 - 1)
#define FX_W2(j) (RING ? (u + K3 * K - PR2 + 1 + (j)) % K3 : (j))

                // ---- conv1 (valid) at t1 = s
                if (s < L1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        int raw;
                        if (!AHEAD) raw = seq_byte(nt, s + K - 1);
                        else {
                            // oldest entry of the look-ahead queue; its slot takes the byte PF positions further on
                            const int slot = RING ? u % PF : 0;
                            raw = rq[slot][nt];
                            if (!RING) {
#pragma unroll
                                for (int q = 0; q + 1 < PF; ++q) rq[q][nt] = rq[q + 1][nt];
                            }
                            if (s + K - 1 + PF < L) rq[RING ? slot : PF - 1][nt] = seq_byte(nt, s + K - 1 + PF);
                        }
                        int c = lut_s[raw];
                        if (c == 0xFF) { bad = true; c = 0; }
                        cw[FX_CW(K - 1)][nt] = c;
                    }
                }
                if (s < L1 && act) {
                    f4 o1[FT][NT];
                    init_bias<FT, NT>(cb, o1, g);
                    if (G1) {
                        // one-hot conv == sum of K kernel rows selected by the codes: LDS gather + VALU adds
                        // (takes the 2*K*A*F "multiply by one-hot" FLOP per position off the MFMA pipe)
#pragma unroll
                        for (int j = 0; j < K; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const float* rowp = w1p + (j * A + cw[FX_CW(j)][nt]) * FX_C1_ROW(FT) + 4 * g;
#pragma unroll
                                for (int mo = 0; mo < FT; ++mo) {
                                    const f4 w = *reinterpret_cast<const f4*>(rowp + 16 * mo);
                                    o1[mo][nt] += w;
                                }
                            }
                    } else {
#pragma unroll
                        for (int st = 0; st < S1; ++st) {
                            const int j = (4 * st) / A;          // tap (compile-time after unroll)
                            const int a0 = (4 * st) % A;         // first alphabet index of the step
                            const int sg = st >> 2, r = st & 3;
                            float b[NT];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) b[nt] = (cw[FX_CW(j)][nt] == a0 + g) ? 1.f : 0.f;
#pragma unroll
                            for (int mo = 0; mo < FT; ++mo) {
                                const float a = reinterpret_cast<const float*>(&w_first[(sg * FT + mo) * 64 + lane])[r];
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) o1[mo][nt] = mfma16(a, b[nt], o1[mo][nt]);
                            }
                        }
                    }
                    relu_tiles<FT, NT>(o1);
#pragma unroll
                    for (int t = 0; t < FT; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) win1[FX_W1NEW][t][nt] = o1[t][nt];
                } else {
#pragma unroll
                    for (int t = 0; t < FT; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) win1[FX_W1NEW][t][nt] = splat4(0.f);
                }

                // ---- conv2 (same) at t2 = s - PR2; tap j reads out1[t2 + j - PL2] = out1[s - (K-1) + j]
                const int t2 = s - PR2;
                if (t2 >= 0 && t2 < L1 && act) {
                    f4 o2[FT][NT];
                    init_bias<FT, NT>(cb + 16 * FT, o2, g);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int pp = t2 + j - PL2;
                        if (pp >= 0 && pp < L1)            // zero padding contributes nothing
                            mma_layer<FT, FT, NT, PRIO>(w_c2 + j * FT * FT * 64, win1[FX_W1(j)], o2, lane);
                    }
                    relu_tiles<FT, NT>(o2);
#pragma unroll
                    for (int t = 0; t < FT; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) win2[FX_W2NEW][t][nt] = o2[t][nt];
                } else {
#pragma unroll
                    for (int t = 0; t < FT; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) win2[FX_W2NEW][t][nt] = splat4(0.f);
                }

                // ---- conv3 (same, kernel A-1) at t3 = t2 - PR3; tap j reads out2[t3 + j - PL3] = out2[t2 - (K3-1) + j]
                //      (MaxPooling1D(1) between conv2 and conv3 is the identity, cnn.py:40)
                const int t3 = t2 - PR3;
                if (t3 >= seg_lo && t3 < seg_hi) {
                    f4 o3[FT][NT];
                    init_bias<FT, NT>(cb + 32 * FT, o3, g);
#pragma unroll
                    for (int j = 0; j < K3; ++j) {
                        const int pp = t3 + j - PL3;
                        if (pp >= 0 && pp < L1)
                            mma_layer<FT, FT, NT, PRIO>(w_c3 + j * FT * FT * 64, win2[FX_W2(j)], o3, lane);
                    }
                    // GlobalMaxPooling1D of relu(o3): gmax starts at 0
#pragma unroll
                    for (int t = 0; t < FT; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) gmax[t][nt] = pool_max4(gmax[t][nt], o3[t][nt]);
                }
#undef FX_CW
#undef FX_W1NEW
#undef FX_W1
#undef FX_W2NEW
#undef FX_W2
                }
            }
            }

            if (SEG) {
                // ---- segment maxima -> LDS; wave 0 folds them and carries on with the dense head
                f4* seg_slot = reinterpret_cast<f4*>(smem + lds_floats + 64 + 12);  // after the LUT and the counters
                __syncthreads();                                  // previous tile's readers are done
#pragma unroll
                for (int t = 0; t < FT; ++t) seg_slot[((tid >> 6) * FT + t) * 64 + lane] = gmax[t][0];
                __syncthreads();
                if (tid >= 64) continue;
#pragma unroll
                for (int w = 1; w < WAVES; ++w)
#pragma unroll
                    for (int t = 0; t < FT; ++t) gmax[t][0] = pool_max4(gmax[t][0], seg_slot[(w * FT + t) * 64 + lane]);
                if constexpr (HEAD) {
                    if (seg_sb > 1) {
                        // ---- the workgroups of this tile meet in a zeroed global pool: atomicMax on the float bits (relu
                        //      outputs are >= +0, where float order == unsigned order); the last to arrive runs the head
                        const int64_t unit = (int64_t)m * p.TG + tg;
                        unsigned* pl = p.seg_pool + (unit * FT * 64 + lane) * 4;
#pragma unroll
                        for (int t = 0; t < FT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) atomicMax(&pl[t * 256 + r], __float_as_uint(gmax[t][0][r]));
                        __threadfence();
                        unsigned ticket = 0;
                        if (lane == 0) ticket = atomicAdd(&p.seg_cnt[unit], 1u);
                        ticket = __builtin_amdgcn_readfirstlane(ticket);
                        if (ticket != (unsigned)seg_sb - 1u) continue;
                        __threadfence();
#pragma unroll
                        for (int t = 0; t < FT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {            // device-coherent reads; the entry goes back to zero
                                gmax[t][0][r] = __uint_as_float(__hip_atomic_load(&pl[t * 256 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                __hip_atomic_store(&pl[t * 256 + r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        if (lane == 0) __hip_atomic_store(&p.seg_cnt[unit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if constexpr (!HEAD) {
                f4* dst = p.pool_out + (((int64_t)m * p.TG + tg) * FT) * 64 + lane;
#pragma unroll
                for (int t = 0; t < FT; ++t) dst[t * 64] = gmax[t][0];
                continue;
            }
            // ---- dense head: F -> H relu -> H relu -> (dropout inactive) -> 1
            FX_PHASE_STAMP(8);
            if (CAN_STAGE && !head_ready) {
                // staged fill: the head's weights are brought by the idle waves; wait for their chunk count
                while (__hip_atomic_load(&stage_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n_chunks)
                    __builtin_amdgcn_s_sleep(1);
                __threadfence_block();
                head_ready = true;
            }
            asm volatile("" ::: "memory");
            if (!DENSE_LDS) {
                // weights streamed from L2: launder the base pointer per tile, otherwise LICM hoists the
                // 64-bit address of every 1 KiB block out of the tile loop and spills them
                asm volatile("" : "+v"(w_d1), "+v"(w_d2), "+v"(db));
            }
            f4 h1[HT][NT], h2[HT][NT];
            init_bias<HT, NT>(db, h1, g);
            mma_layer<FT, HT, NT, PRIO>(w_d1, gmax, h1, lane);
            relu_tiles<HT, NT>(h1);
            FX_PHASE_STAMP(9);
            init_bias<HT, NT>(db + 16 * HT, h2, g);
            mma_layer<HT, HT, NT, PRIO>(w_d2, h1, h2, lane, p.rlh);
            relu_tiles<HT, NT>(h2);
            float y[NT];
            final_dot<HT, NT>(db + 32 * HT, db[48 * HT], h2, y, g);
#if defined(FX_AB)
            if (NT == 1 && !SEG && FX_FM_ON(p)) fx_fused_mean_tile(p.out, p.out_sm, p.M, m, tg, n[0], p.N, fx_nan_to_num(y[0]), g == 0, p.fm_mean, p.fm_cnt, lane);
            else
#endif
            if (g == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (n[nt] < p.N) p.out[n[nt] * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[nt]);
            }
            FX_TILE_DONE();
        }
        if constexpr (QT) {
            if (nq) {                                             // (workgroup-uniform)
                // quad j = waves 4j .. 4j + 3 (hardware places consecutive waves on consecutive SIMDs; nothing depends on it), roles
                // rotated from quad to quad as in score_cnn_quad.hip; the exchange buffers lie behind the LUT and the counters
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), quad = wv >> 2, q = ((wv & 3) + quad) & 3;
                const bool qlive = quad < nq;
                f4* xq = reinterpret_cast<f4*>(smem + lds_floats + 64 + 12) + (qlive ? quad : 0) * (2 * QXT * 64);
                const int64_t qtg = t_main + (qlive ? quad : 0);
                const int64_t qn = qtg * 16 + sq;
                const uint8_t* qrow = p.ascii + (qn < p.N ? qn : 0) * L;
                float yq = 0.f;
                fx_cnn_quad_round<FT, HT, K, L1S, QXT>(qlive, q, lane, qrow, lut_s, w1p, cb, w_c2, w_c3, w_d1, w_d2, db, p.rlh, xq, xq + QXT * 64, bad, yq);
#if defined(FX_AB)
                if (FX_FM_ON(p)) {
                    if (qlive && q == 0) fx_fused_mean_tile(p.out, p.out_sm, p.M, m, qtg, qn, p.N, fx_nan_to_num(yq), g == 0, p.fm_mean, p.fm_cnt, lane);
                } else
#endif
                if (qlive && q == 0 && g == 0 && qn < p.N) p.out[qn * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(yq);
            }
        }
        if (CAN_STAGE && staged && !got_tile) {
            // this wave has no tile: bring the dense head's weights, 8 KiB per chunk
            const f4* src = reinterpret_cast<const f4*>(p.w[m]) + p.conv_floats / 4;
            f4* dst = reinterpret_cast<f4*>(smem) + p.conv_floats / 4;
            int mine = 0;
            for (;;) {
                int c = 0;
                if (lane == 0) c = atomicAdd(&stage_ctl[0], 1);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= n_chunks) break;
                f4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = c * 512 + k * 64 + lane;
                    if (i < dense_f4) v[k] = src[i];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = c * 512 + k * 64 + lane;
                    if (i < dense_f4) dst[i] = v[k];
                }
                ++mine;
            }
            __threadfence_block();                        // this wave's LDS stores have landed
            if (lane == 0 && mine) atomicAdd(&stage_ctl[1], mine);
        }
    }
    fx_stamp(p.trace, 6);
    if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

template <int A, int K, int FT, int HT, int NT, bool DENSE_LDS, int WAVES, bool G1, int L1S = 0, bool PRIO = false,
          bool SEG = false, bool HEAD = true, bool STG = false, bool QT = false>
int launch_g(fx_engine* e, const CnnArgs& a_in, size_t lds_bytes) {
    constexpr int waves = WAVES;
    auto kern = k_score_cnn_mfma<A, K, FT, HT, NT, DENSE_LDS, WAVES, G1, L1S, PRIO, SEG, HEAD, STG, QT>;
    if (QT) lds_bytes += (size_t)3 * 2 * 8 * 1024;               // three quads' exchange buffers (2 x 8 KiB each)
    if (SEG) lds_bytes += (size_t)WAVES * FT * 64 * 16;          // segment-maxima slots
    const bool ahead = STG && e->cnn_stage_host >= 2 && lds_bytes + 2 * (size_t)WAVES * ((16 * (size_t)a_in.L + 15) / 16 * 16) <= (size_t)e->max_lds;
    if (STG) lds_bytes += (size_t)WAVES * ((16 * (size_t)a_in.L + 15) / 16 * 16) * (ahead ? 2 : 1);   // a tile's bytes per wave (two scratches with the look-ahead)
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    CnnArgs a = a_in;
#if defined(FX_AB)
    if (SEG || !HEAD || NT != 1) a.fm_mean = nullptr;            // (forms that share a tile among waves / leave the head to another kernel: the mean kernel follows)
#endif
    a.stw_stride = STG ? (int)((16 * (size_t)a.L + 15) / 16 * 16) : 0;
    a.stw_ahead = ahead ? 1 : 0;
    if (e->rows_req.on) {
        // (a launched-first call: this kernel waits for its rows tile by tile -- the forms that share a tile among waves do not)
        if (SEG || NT != 1 || !HEAD || e->rows_req.relay.flags) return FX_EUNSUPPORTED;   // (no relay form of this kernel)
        if (!fx_rows_plan(e)) return FX_EUNSUPPORTED;
        a.ready = e->rows_req.r;
        e->rows_req.used = true;
    }
    int64_t U = (int64_t)a.M * a.TG;
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    // never more workgroups than work units
    int64_t need = U;                                // small batches: one unit per workgroup (lowest latency)
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (SEG && HEAD && a.seg_sb > 1) blocks = U * a.seg_sb;       // multi-workgroup segment form: exactly units x seg_sb
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(waves * 64), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
#if defined(FX_AB)
    if (a.fm_mean) e->fused_mean_done = true;
#endif
    return FX_OK;
}

template <int A, int K, int FT, int HT, int NT, bool DENSE_LDS, int WAVES>
int launch_inst(fx_engine* e, const CnnArgs& a, size_t lds_bytes) {
#if defined(FX_AB)   // (the MFMA form of the one-hot conv1 -- 3-15 % slower than the LDS gather -- lives in the A/B build, `make ab`)
    if (e->cnn_conv1_mfma) return launch_g<A, K, FT, HT, NT, DENSE_LDS, WAVES, false>(e, a, lds_bytes);
#endif
    return launch_g<A, K, FT, HT, NT, DENSE_LDS, WAVES, true>(e, a, lds_bytes);
}

}  // namespace
