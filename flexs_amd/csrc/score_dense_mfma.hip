// K2 score_mlp / score_ge: string -> one-hot -> Dense stack, fused, on f32 MFMA.
//
// Replaces keras_model.py:69-79 + mlp.py:21-31 / global_epistasis_model.py:26-36.
// Same transposed-MFMA formulation and work split as score_cnn_mfma.hip (see
// mfma_common.h): a wave owns NT tiles of 16 sequences, activations stay in
// accumulator registers from layer to layer, the member's weights sit in LDS.
//   MLP  layer 1 is a one-hot MFMA contraction over k = l*A + a (k-step = 4 rows),
//        layers 2-3 are HxH MFMA, layer 4 a per-lane dot + 2 cross-lane adds.
//   GE   layer 1 (L*A -> 1) is a per-lane gather-sum from an LDS table, layer 2
//        (1 -> H) a VALU fma written directly in B-operand layout, layer 3 HxH
//        MFMA, layer 4 the dot.
// Algorithmic work per sequence per member: L + 4 bytes; 2*MACs FLOP with
// MACs = L*A*H + 2*H*H + H (MLP) or L*A + H + H*H + H (GE).
#include <cstring>

#include "fx_common.h"
#include "mfma_common.h"
// SLAB: input tiles per slab (3 and 4 measured no faster once the slabs go global -> LDS directly: profiles/r6_slab_dma_ab.log)
#define FX_SLAB_KG 2
#include "score_dense_tile.h"
#include "score_dense_l1.h"

namespace {

struct DenseArgs {
    const uint8_t* ascii;
    const uint8_t* lut;
    const float* w[FX_MAX_M];
    float* out;
    unsigned* err;
    unsigned long long* trace;  // in-kernel timeline (null = off), see fx_stamp
    int wave_prio;              // 1 = fx_stagger_priority
    int64_t N, TG;
    int M, Mtot, m_off;
    int64_t out_sn, out_sm;     // out[n * out_sn + column * out_sm]
    int L, A, rlh;
    int SG1, off_first, off_w1p, off_d2, off_d3, off_db, total_floats;
    int lds_from, lds_floats;   // the LDS image is packed[lds_from .. lds_from + lds_floats)
    int off_w1pair, pair_floats; // PAIR (MLP, 4 letters): pre-summed first-layer rows per pair of positions, staged after the image
    int stage_stride;           // > 0: bytes of LDS scratch per wave for the tile's sequence bytes (fx_stage_tile); 0 = read them from global memory
    // BT (GlobalEpistasis): first layer as a per-position table indexed by (raw byte - bt_base), Lpad x 32 floats at LDS offset 0
    const float* bt[FX_MAX_M];
    int Lpad;                   // L rounded up to 32 positions (the padding rows are zeros)
    int bt_base;                // smallest byte of the alphabet; every letter lies in [bt_base, bt_base + 32)
    int validate;               // BT: 1 = this launch checks the characters, each tile by ONE of its members (once per call, not once per member)
    // the tiles of a workgroup that do not divide among its four SIMDs are walked by groups of 8 waves (score_dense_tile.h):
    // 0 = off, else the number of groups that find room for their exchange buffers (PAIR: over the pair rows; BT: at coop_off)
    int coop, coop_off;
    // SLAB: the last (tiles mod 8) tiles of a workgroup, up to this many, are walked one at a time by the workgroup's 8 waves together
    // (fx_dense_tile8, the H x H blocks straight from L2) instead of costing a whole lockstep round; 0 = off
    int slab_coop;
    const f4* h1;               // W1G: the relu'd first-layer sums, [(member * TG + tile) * HT + mo][64 lanes], computed by k_mlp_l1_pos (score_dense_l1.h); nullptr = gather here
    FxRowsReady ready;          // launched-first host call: the rows arrive while the kernel runs (words == nullptr: they are all there)
    FxRelay relay;              // ... and member 0's workgroups pass them on to the other members through device memory (flags == nullptr: no)
};

// GE first layer as a table indexed by the RAW byte: tab[l][b - base] = w1[l * A + lut[b]] for the 32 byte values from
// `base` on (every FLEXS alphabet spans fewer than 32 code points), 0 for bytes outside the alphabet and for the padding
// rows l >= L.  Built once per (weights, LUT) by the launcher.
__global__ void k_ge_bytetab(const float* __restrict__ w1, const uint8_t* __restrict__ lut, int L, int A, int Lpad, int base,
                             float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Lpad * 32) return;
    const int l = idx >> 5, b = base + (idx & 31);
    const int c = b < 256 ? lut[b] : 0xFF;
    out[idx] = (l < L && c < A) ? w1[l * A + c] : 0.f;
}

// SLAB: one H x H layer of a lockstep round; WNEXT = the layer whose first slab is asked for during this one's last (nullptr: none follows)
#define FX_SLAB_LAYER(W, WNEXT, IN, OUT) mma_layer_slab_dma<HT, HT, KG, WAVES>(W, WNEXT, slab, IN, OUT, lane, p.rlh, slab_st)
// SLAB (with DG): the HxH blocks are streamed through LDS once per round of WAVES tiles (mma_layer_slab) instead
// of once per tile per wave; the waves of a workgroup then walk the tiles in lockstep.
// PF (rows in HOST memory, the PAIR / BT forms that copy a tile's bytes into LDS): a wave claims its NEXT tile before it scores the
// current one and asks for that tile's bytes straight into a second LDS scratch (fx_stage_tile_dma) -- the PCIe round trip, 3 us
// idle and ~16 us with the link saturated, then lies beside a tile's work instead of in front of every tile of a latency-bound walk.
template <int KIND, int A, int HT, int NT, int WAVES, bool G1, bool W1G, bool DG, bool SLAB = false, bool BT = false, bool PAIR = false, bool PF = false>
__global__ void __launch_bounds__(WAVES * 64) k_score_dense_mfma(DenseArgs p) {
    static_assert(!PF || ((PAIR || BT) && NT == 1 && !SLAB), "PF is a form of the staged-tile first layers");
    static_assert(!PAIR || (KIND == FX_MLP && G1 && !W1G && (!DG || SLAB) && NT == 1), "PAIR is the MLP gather form on a 4-letter alphabet (H x H blocks in LDS, or streamed through slabs)");
    static_assert(!SLAB || (DG && NT == 1), "SLAB streams the L2-resident blocks of the one-tile form");
    static_assert(!BT || (KIND == FX_GE && NT == 1 && !DG && !SLAB), "BT is the GlobalEpistasis byte-table form");
    constexpr int KG = FX_SLAB_KG;                              // input tiles per slab
    // the last (tiles mod 4) tiles of a workgroup shared by wave groups instead of making one SIMD run an extra tile
    constexpr bool COOP_OK = NT == 1 && !SLAB && !DG && !W1G && WAVES == 16 && (PAIR || BT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L;
    float* img = smem + (BT ? p.Lpad * 32 : 0);                          // weight image (after the byte table, if any)
    float* wpair = img + p.lds_floats;                                   // PAIR: pair rows right after the image
    float* aux = wpair + (PAIR ? p.pair_floats : 0);
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(aux);
    int* next_tile = reinterpret_cast<int*>(aux + 64);                  // 4 work counters (one per SIMD), after the 256-byte LUT
    int* simd_waves = next_tile + 4;                                    // 4 wave counts (workgroup's waves per SIMD)
    f4* slab = reinterpret_cast<f4*>(aux + 64 + 8);                     // SLAB: 2 x KG*HT KiB
    uint8_t* stw = reinterpret_cast<uint8_t*>(aux + 64 + 8) + (tid >> 6) * p.stage_stride * (PF ? 2 : 1);   // this wave's sequence-byte scratch (never with SLAB)
    [[maybe_unused]] uint8_t* stw_alt = stw + p.stage_stride;           // PF: the scratch the next tile's bytes arrive in
    fx_stamp(p.trace, 0);
    if (p.wave_prio) fx_stagger_priority();
    const int simd = fx_simd_id();
    fx_stamp(p.trace, 7, (unsigned long long)simd + 1);
    for (int i = tid; i < 64; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];
    if (tid < 4) simd_waves[tid] = 0;

    int64_t u_lo, u_hi;
    fx_unit_range(p.TG, p.M, u_lo, u_hi, p.relay.flags != nullptr && p.relay.spread);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    bool bad = false;
    [[maybe_unused]] unsigned tiles_done = 0;
    FxSimdShare share{0, 1, 1};
    if (BT && p.stage_stride && lane < 32) stw[16 * L + lane] = (uint8_t)p.bt_base;   // what the padded trips of the last row read
    if (PF && BT && p.stage_stride && lane < 32) stw_alt[16 * L + lane] = (uint8_t)p.bt_base;

    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();
        if (tid < 4) next_tile[tid] = 0;
        if (m == m_first) fx_count_simd_wave(simd_waves, simd);      // (zeroed before the barrier above)
        {
            const f4* src = reinterpret_cast<const f4*>(p.w[m] + p.lds_from);
            f4* dst = reinterpret_cast<f4*>(img);
            // (both segments in one pass with 12 loads in flight per thread measured slower: the fill is bound by the
            // request rate of 256 workgroups asking at once, not by round trips -- profiles/r2 trace)
            fill_lds(dst, src, p.lds_floats / 4);
            if (BT) fill_lds(reinterpret_cast<f4*>(smem), reinterpret_cast<const f4*>(p.bt[m]), p.Lpad * 8);
            if (PAIR) fill_lds(reinterpret_cast<f4*>(wpair), reinterpret_cast<const f4*>(p.w[m] + p.off_w1pair), p.pair_floats / 4);
        }
        __syncthreads();
        if (m == m_first) share = fx_simd_share(simd_waves, simd);
        if (m == m_first) fx_stamp(p.trace, 1);
        // W1G: the (large) first-layer rows stay in global memory / L2, only the HxH blocks sit in LDS
        const float* w_first = W1G ? p.w[m] + p.off_first : img + (p.off_first - p.lds_from);
        const float* w1p = W1G ? p.w[m] + p.off_w1p : img + (p.off_w1p - p.lds_from);
        // DG: HxH blocks too large for LDS (H > 128) stream from L2; the LDS image then ends before them
        const f4* w_d2 = reinterpret_cast<const f4*>(DG ? p.w[m] + p.off_d2 : img + (p.off_d2 - p.lds_from));
        const f4* w_d3 = reinterpret_cast<const f4*>(DG ? p.w[m] + p.off_d3 : img + (p.off_d3 - p.lds_from));
        const float* db = DG ? p.w[m] + p.off_db : img + (p.off_db - p.lds_from);

        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;

        // 25 tiles on four SIMDs are 7 + 6 + 6 + 6: the SIMD with the odd tile sets the time of the workgroup (at 1e5
        // sequences every tile is 1 / 6 of the launch).  The tiles that do not divide by four are left out of the shares and
        // walked afterwards by groups of 8 waves, output tiles dealt to the waves (~1.5 us for two tiles at once instead of
        // ~6 us for a lone wave's tile).  PAIR: the groups' exchange buffers lie over the pair rows, which the first layer
        // of that one round still reads -- so all shared tiles must fit one round, else none is shared.
        int ncoop = 0;
        if constexpr (COOP_OK) {
            const int64_t cnt = t_hi - t_lo;
            // (only where the odd tile weighs: up to 8 tiles per SIMD.  The shared walk is a once-per-workgroup code path --
            //  cold in the instruction cache, five barriers -- and costs ~4 us against ~6 us for a lone wave's tile: -5 % at
            //  5 and -2..-4 % at 6 tiles per SIMD, nothing at 60, +1 % when a workgroup's range spans three members)
            if (p.coop && cnt >= 4 && u_hi - u_lo <= 32 && !p.ready.words) {   // (rows that arrive during the run: no shared walk -- its bytes are asked for up front)
                const int r = (int)(cnt & 3);
                ncoop = (PAIR && r > p.coop) ? 0 : r;
            }
        }
        // SLAB (H > 128): a lockstep round streams both H x H layers (2 x 169 KiB at H = 200) through the slabs whether 8 tiles are
        // live or 1 -- 52 us either way at H = 200 -- and 1e5 sequences are 24.4 tiles per workgroup: three rounds for 150 workgroups,
        // a fourth round with ONE live tile for the other 106, which set the launch (profiles/r6_trace_probe.json: exits at 157 and
        // 214 us).  Up to slab_coop leftover tiles are instead walked by the 8 waves together, output tiles dealt to the waves.
        int nslab = 0;
        if constexpr (SLAB && NT == 1 && WAVES == 8) {
            const int r = (int)((t_hi - t_lo) % WAVES);
            if (r >= 1 && r <= p.slab_coop) nslab = r;
        }
        const int64_t t_end = t_hi - nslab;                     // SLAB: end of the lockstep rounds
        const int64_t t_main = t_hi - ncoop;
        // the bytes of the shared tile this wave's group will walk, requested now (a full tile of <= 1 KiB: 16 bytes per lane),
        // so that the round after the main loop does not start with a trip to memory
        [[maybe_unused]] FxBytes16 coop_pre{};
        [[maybe_unused]] bool coop_pre_ok = false;
        if constexpr (COOP_OK) {
            const int grp0 = (tid >> 6) >> 3;
            if (ncoop && p.stage_stride && 16 * L <= 1024 && grp0 < p.coop && grp0 < ncoop && (t_main + grp0 + 1) * 16 <= p.N) {
                coop_pre_ok = true;
                if (lane * 16 < 16 * L) coop_pre = *reinterpret_cast<const FxBytes16*>(p.ascii + (t_main + grp0) * 16 * L + lane * 16);
            }
        }
        // the workgroup's tiles in shares per SIMD (proportional to the waves it hosts); the waves of a SIMD pull from
        // their share's counter
        const int64_t s_lo = t_lo + (t_main - t_lo) * share.before / share.total;
        const int64_t s_hi = t_lo + (t_main - t_lo) * (share.before + share.mine) / share.total;
        // launched-first call (as in score_cnn_kernel.h): the share is walked from its first block of stage 0 on, and around
        const bool rows_arrive = !SLAB && NT == 1 && p.ready.words != nullptr;
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
        [[maybe_unused]] int64_t pf_next = -2;              // PF: the tile claimed ahead (-2: none claimed, -1: the share is exhausted)
        [[maybe_unused]] bool pf_issued = false;            // PF: ... and its bytes are on their way into stw_alt
        [[maybe_unused]] FxSlabStream slab_st{0, false};    // SLAB: the stream of slabs across layers and rounds (mma_layer_slab_dma)
        for (int64_t round = 0;; ++round) {
            // SLAB: lockstep rounds of WAVES tiles; waves without a tile in the last round run along on tile 0
            int64_t tg_want = t_lo + round * WAVES + (tid >> 6);
            if (!SLAB) {
                tg_want = (PF && pf_next != -2) ? pf_next : pull_tile();
                if (tg_want < 0) break;
                // (a relay's readers wait for member 0's workgroups, not for the host)
                if (rows_arrive && (!p.relay.flags || m + p.m_off == 0)) fx_rows_wait(p.ready, (int)(tg_want % p.ready.Q), rows_known, p.err);
            }
            if (SLAB && t_lo + round * WAVES >= t_end) break;
            [[maybe_unused]] const bool slab_more = SLAB && t_lo + (round + 1) * WAVES < t_end;   // another lockstep round follows
            const bool live = !SLAB || tg_want < t_end;
            const int64_t tg = live ? tg_want : t_lo;
            if (tiles_done == 0) fx_stamp(p.trace, 2);
            asm volatile("" ::: "memory");               // keep LDS weight reads inside the tile loop
            if (DG) asm volatile("" : "+v"(w_d2), "+v"(w_d3), "+v"(db));   // L2-streamed blocks: no hoisted addresses
            int64_t n[NT];
            const uint8_t* row[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                n[nt] = (tg * NT + nt) * 16 + sq;
                row[nt] = p.ascii + (n[nt] < p.N ? n[nt] : 0) * L;
                if (rows_arrive) row[nt] = p.ascii + tg * p.ready.pitch + (n[nt] < p.N ? sq : 0) * L;   // (tile-pitched staging)
            }
            // the tile's bytes through this wave's LDS scratch (PAIR / BT first layers); lanes past the batch use row 0
            const int64_t tile_rows = p.N - tg * 16 < 16 ? p.N - tg * 16 : 16;
            if (p.stage_stride && (PAIR || BT)) {
                const int64_t tile_pitch = rows_arrive ? (int64_t)p.ready.pitch : (int64_t)16 * L;
                const int64_t at_byte = tg * tile_pitch;
                const bool relayed = rows_arrive && p.relay.flags;
                const bool copier = relayed && m + p.m_off == 0;
                bool have_bytes = false;
                if constexpr (PF) {
                    if (pf_issued) {
                        // this tile's bytes were asked for a tile ago: they are in the other scratch (once the loads have landed)
                        uint8_t* t = stw; stw = stw_alt; stw_alt = t;
                        fx_wait_vm(0);
                        have_bytes = true;
                    }
                    pf_next = -2; pf_issued = false;
                }
                if (have_bytes) {
                    if (copier) {
                        fx_relay_from_lds(stw, (int)tile_rows * L, lane, p.relay.dst + tg * (int64_t)p.relay.pitch);
                        fx_wait_vm(0);
                        if (lane == 0) __hip_atomic_store(p.relay.flags + tg, p.relay.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if (relayed) {
                    const int64_t relay_byte = tg * (int64_t)p.relay.pitch;
                    if (copier) {
                        fx_stage_tile_pass(p.ascii + at_byte, (int)tile_rows * L, stw, lane, p.relay.dst + relay_byte);
                        fx_wait_vm(0);                              // this wave's stores have been taken ...
                        if (lane == 0) __hip_atomic_store(p.relay.flags + tg, p.relay.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... the tile is there
                    } else {
                        fx_relay_wait(p.relay.flags + tg, p.relay.seq, p.err);
                        fx_stage_tile_from(p.relay.dst + relay_byte, (int)tile_rows * L, stw, lane);
                    }
                } else fx_stage_tile(p.ascii + at_byte, (int)tile_rows * L, stw, lane);
                if constexpr (PF) {
                    // claim the next tile now and, if it is whole and its rows are known to be there, ask for its bytes (waves that
                    // read the host staging area: all of them without a relay, member 0's with one)
                    if (!relayed || copier) {
                        pf_next = pull_tile();
                        if (pf_next >= 0 && (pf_next + 1) * 16 <= p.N && (!rows_arrive || (int)(pf_next % p.ready.Q) < rows_known)) {
                            fx_stage_tile_dma(p.ascii + pf_next * tile_pitch, 16 * L, stw_alt, lane);
                            pf_issued = true;
                        }
                    }
                }
            }
            const uint8_t* srow = stw + (n[0] < p.N ? sq : 0) * L;
            f4 h[HT][NT];
            float y[NT];
            if (KIND == FX_MLP) {
                static_assert(KIND != FX_MLP || A % 4 == 0, "one-hot k-steps must not straddle a position");
                // ---- layer 1: relu(b1 + onehot @ W1), contraction index k = l*A + a
                [[maybe_unused]] bool have_h1 = false;
                if constexpr (W1G && NT == 1) {
                    if (p.h1) {
                        // the first layer was taken position-major by k_mlp_l1_pos (relu'd already; the relu below is idempotent)
                        const f4* src = p.h1 + (((int64_t)m * p.TG + tg) * HT) * 64 + lane;
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) h[mo][0] = src[mo * 64];
                        have_h1 = true;
                    }
                }
                if (!have_h1) init_bias<HT, NT>(db, h, g);
                if (have_h1) {
                } else if constexpr (PAIR) {
                    // 4-letter alphabet: one pre-summed row per PAIR of positions (16 letter pairs), i.e. half the LDS
                    // traffic, half the adds and half the address arithmetic of the row-per-position gather below --
                    // the first layer is what a tile waits for while the matrix pipe idles (profiles/archive/r2_trace_probe)
                    unsigned seen1 = 0;
                    const int np2 = L >> 1;
                    auto pair_layer = [&](auto rb) {
                        for (int p0 = 0; p0 < np2; p0 += 2) {
                            asm volatile("" ::: "memory");
                            int raw[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) raw[k] = rb[2 * p0 + k < L ? 2 * p0 + k : 0];     // independent loads
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                if (p0 + k < np2) {
                                    const unsigned c0 = lut_s[raw[2 * k]], c1 = lut_s[raw[2 * k + 1]];
                                    seen1 |= c0 | c1;                  // a code is < 4, or 0xFF: tested once per tile
                                    const unsigned idx = ((c0 & 3u) << 2) | (c1 & 3u);
                                    const float* rowp = wpair + ((p0 + k) * 16 + idx) * (16 * HT + FX_PAIR_PAD) + 4 * g;
#pragma unroll
                                    for (int mo = 0; mo < HT; ++mo) h[mo][0] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                                }
                            }
                        }
                        if (L & 1) {
                            const unsigned c0 = lut_s[rb[L - 1]];
                            seen1 |= c0;
                            const float* rowp = wpair + (np2 * 16 + (c0 & 3u)) * (16 * HT + FX_PAIR_PAD) + 4 * g;
#pragma unroll
                            for (int mo = 0; mo < HT; ++mo) h[mo][0] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                        }
                    };
                    if (p.stage_stride) pair_layer((fx_lds_u8p)srow);
                    else pair_layer(row[0]);
                    bad |= seen1 >= 0x80u;
                } else if (G1) {
                    // one-hot layer == sum of L kernel rows selected by the codes: LDS gather + VALU adds
                    unsigned seen1 = 0;
                    const unsigned amax1 = (unsigned)p.A - 1u;
                    for (int l0 = 0; l0 < L; l0 += 4) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            int raw[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) raw[k] = row[nt][l0 + k < L ? l0 + k : 0];   // independent loads
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int l = l0 + k;
                                if (l < L) {
                                    const unsigned c = lut_s[raw[k]];
                                    seen1 |= c;                       // a code is < A <= 127, or 0xFF: tested once per tile
                                    const unsigned ci = c < amax1 ? c : amax1;
                                    const float* rowp = w1p + (l * p.A + ci) * (16 * HT) + 4 * g;
#pragma unroll
                                    for (int mo = 0; mo < HT; ++mo) h[mo][nt] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                                }
                            }
                        }
                    }
                    bad |= seen1 >= 0x80u;
                } else {
                    for (int sg = 0; sg < p.SG1; ++sg) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k0 = 16 * sg + 4 * r;            // first row of this k-step
                            const int l = k0 / A, a0 = k0 % A;
                            float b[NT];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                int c = 0xFE;
                                if (l < L) {
                                    c = lut_s[row[nt][l]];
                                    bad |= (c == 0xFF);
                                }
                                b[nt] = (c == a0 + g) ? 1.f : 0.f;
                            }
#pragma unroll
                            for (int mo = 0; mo < HT; ++mo) {
                                const float a = w_first[((sg * HT + mo) * 64 + lane) * 4 + r];
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) h[mo][nt] = mfma16(a, b[nt], h[mo][nt]);
                            }
                        }
                    }
                }
                relu_tiles<HT, NT>(h);
                FX_PHASE_STAMP(8);
                // ---- layers 2, 3
                f4 h2[HT][NT];
                init_bias<HT, NT>(db + 16 * HT, h2, g);
                if constexpr (SLAB) FX_SLAB_LAYER(w_d2, w_d3, h, h2);
                else mma_layer<HT, HT, NT>(w_d2, h, h2, lane, p.rlh);
                relu_tiles<HT, NT>(h2);
                FX_PHASE_STAMP(9);
                asm volatile("" ::: "memory");
                init_bias<HT, NT>(db + 32 * HT, h, g);
                if constexpr (SLAB) FX_SLAB_LAYER(w_d3, slab_more ? w_d2 : nullptr, h2, h);
                else mma_layer<HT, HT, NT>(w_d3, h2, h, lane, p.rlh);
                relu_tiles<HT, NT>(h);
                FX_PHASE_STAMP(10);
                final_dot<HT, NT>(db + 48 * HT, db[64 * HT], h, y, g);
            } else {
                // ---- GE layer 1: s = relu(b1 + sum_l w1[l*A + code_l])   (scalar per sequence)
                // (the four lane groups of a sequence each take every 4th position, then two cross-lane adds)
                float s[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) s[nt] = 0.f;
                if constexpr (BT) {
                    // Byte-indexed table at LDS offset 0 (128 bytes per position, index = byte - bt_base folded into the
                    // lane's table pointer): a position costs one address op and one add -- no LUT hop, no clamp, no
                    // per-position selects; every trip covers 8 positions of the lane group (l = g + 4k) through
                    // immediate offsets.  Rows l >= L of the table are zeros, so whole trips run to Lpad; the bytes
                    // read there belong to the next sequences, which exist for every tile but the last ones of the
                    // batch (those take the guarded loop).  VALU instructions cost matrix-pipe time on gfx950: this is
                    // 2 per position instead of ~7.  Characters outside the alphabet are checked
                    // through the LUT by ONE member's unit per tile (all members read the same bytes): a byte outside
                    // [bt_base, bt_base + 32) reads a neighbouring row or nothing, and the call fails anyway.
                    const char* tb = reinterpret_cast<const char*>(smem) + g * 128 - p.bt_base * 4;
                    // padded trips read past the row: into the following rows of the batch (global memory) or of the
                    // staged tile (+ 32 filler bytes behind its last row), which exist for every full tile
                    const bool safe = p.stage_stride ? tile_rows == 16 : (tg * 16 + 16) * (int64_t)L + 32 <= p.N * (int64_t)L;
                    const bool check = p.validate && (int)(tg % p.M) == m;
                    unsigned seen = 0;
                    auto table_layer = [&](auto rp) {
                        if (safe) {
                            // (one trip = 8 bytes per lane in flight; 32 in flight measured slower: registers spill)
                            for (int t = 0; t < p.Lpad; t += 32) {
                                int raw[8];
#pragma unroll
                                for (int k = 0; k < 8; ++k) raw[k] = rp[t + 4 * k];
#pragma unroll
                                for (int k = 0; k < 8; ++k)
                                    s[0] += *reinterpret_cast<const float*>(tb + t * 128 + k * 512 + raw[k] * 4);
                                if (check) {
#pragma unroll
                                    for (int k = 0; k < 8; ++k) {
                                        const unsigned c = lut_s[raw[k]];
                                        seen |= (t + 4 * k + g < L) ? c : 0u;
                                    }
                                }
                            }
                        } else {
                            for (int l = g; l < L; l += 4) {
                                const int raw = rp[l - g];
                                s[0] += *reinterpret_cast<const float*>(tb + (l - g) * 128 + raw * 4);
                                seen |= lut_s[raw];
                            }
                            if (!check) seen = 0;
                        }
                    };
                    if (p.stage_stride) table_layer((fx_lds_u8p)(srow + g));
                    else table_layer(row[0] + g);
                    bad |= seen >= 0x80u;
                } else {
                // eight positions per trip: the byte loads, LUT reads and table reads of a trip are independent,
                // so their latencies overlap instead of chaining.  VALU instructions cost matrix-pipe time on gfx950
                // (DESIGN.md section 4), so the trips that lie entirely inside the sequence run without the `l < L`
                // selects, and bad characters are detected once per tile from the OR of all codes (a code is < A <= 127
                // or 0xFF) instead of a compare + select per position.
                unsigned seen = 0;
                const int nfull = L >= 32 ? (L - 32) / 32 + 1 : 0;       // trips with l0 + 28 < L for every lane group
                const unsigned amax = (unsigned)p.A - 1u;
                for (int t = 0; t < nfull; ++t) {
                    const int l0 = g + 32 * t;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        int raw[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) raw[k] = row[nt][l0 + 4 * k];
                        const float* tab = w_first + l0 * p.A;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned c = lut_s[raw[k]];
                            seen |= c;
                            const unsigned ci = c < amax ? c : amax;   // keeps the read inside the table for a bad character
                            s[nt] += tab[4 * k * p.A + ci];
                        }
                    }
                }
                for (int l0 = g + 32 * nfull; l0 < L; l0 += 16) {        // guarded remainder, four positions per trip
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        int raw[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int l = l0 + 4 * k;
                            raw[k] = row[nt][l < L ? l : 0];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int l = l0 + 4 * k;
                            unsigned c = lut_s[raw[k]];
                            seen |= (l < L) ? c : 0u;
                            const unsigned ci = c < amax ? c : amax;
                            const float w = w_first[(l < L ? l : 0) * p.A + ci];
                            s[nt] += (l < L) ? w : 0.f;
                        }
                    }
                }
                bad |= seen >= 0x80u;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    s[nt] += __shfl_xor(s[nt], 16);
                    s[nt] += __shfl_xor(s[nt], 32);
                    s[nt] += db[0];
                }
                // ---- layer 2: h[ch] = relu(b2[ch] + s * w2[ch]) directly in B-operand layout
                f4 h2[HT][NT];
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) {
                    const f4 w2 = *reinterpret_cast<const f4*>(&db[4 + 16 * mo + 4 * g]);
                    const f4 b2 = *reinterpret_cast<const f4*>(&db[4 + 16 * HT + 16 * mo + 4 * g]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float sv = relu1(s[nt]);
                        f4 v;
                        v.x = relu1(fmaf(sv, w2.x, b2.x));
                        v.y = relu1(fmaf(sv, w2.y, b2.y));
                        v.z = relu1(fmaf(sv, w2.z, b2.z));
                        v.w = relu1(fmaf(sv, w2.w, b2.w));
                        h2[mo][nt] = v;
                    }
                }
                FX_PHASE_STAMP(8);
                // ---- layer 3 (HxH MFMA), layer 4 (dot)
                init_bias<HT, NT>(db + 4 + 32 * HT, h, g);
                if constexpr (SLAB) FX_SLAB_LAYER(w_d3, slab_more ? w_d3 : nullptr, h2, h);
                else mma_layer<HT, HT, NT>(w_d3, h2, h, lane, p.rlh);
                relu_tiles<HT, NT>(h);
                final_dot<HT, NT>(db + 4 + 48 * HT, db[4 + 64 * HT], h, y, g);
            }
            if (g == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (live && n[nt] < p.N) p.out[n[nt] * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[nt]);
            }
            FX_TILE_DONE();
        }
        if constexpr (SLAB && NT == 1 && WAVES == 8) {
            if (nslab) {
                __syncthreads();                                     // every wave is through with the last slab: it becomes the exchange buffer
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                f4* hx = slab;                                       // 2 x HT KiB of the 4 x HT KiB slab area
                for (int i = 0; i < nslab; ++i) {
                    const int64_t tg = t_end + i;
                    const int64_t nn = tg * 16 + sq;
                    const uint8_t* crow = p.ascii + (nn < p.N ? nn : 0) * L;
                    float yc = 0.f;
                    const float* l1 = PAIR ? wpair : nullptr;
                    int l1_form = PAIR ? 1 : 0;
                    if constexpr (W1G) {
                        if (p.h1) { l1 = reinterpret_cast<const float*>(p.h1 + (((int64_t)m * p.TG + tg) * HT) * 64); l1_form = 2; }
                    }
                    fx_dense_tile8<KIND, HT, false>(true, wv, lane, crow, L, p.A, p.rlh, l1_form, w_first, w1p, l1, w_d2, w_d3, db, lut_s, hx, nullptr, bad, yc);
                    if (wv == 0 && g == 0 && nn < p.N) p.out[nn * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(yc);
                    if (i + 1 < nslab) __syncthreads();              // the next tile reuses the exchange buffers
                }
            }
        }
        if constexpr (COOP_OK) {
            if (ncoop) {
                __syncthreads();                                     // every wave is through with its own tiles (and its byte scratch)
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wv >> 3, w8 = wv & 7;
                f4* hx = reinterpret_cast<f4*>(PAIR ? wpair : smem + p.coop_off) + grp * (2 * HT * 64);
                for (int c0 = 0; c0 < ncoop; c0 += p.coop) {
                    const int i = c0 + grp;
                    const bool live = grp < p.coop && i < ncoop;
                    const int64_t tg = t_main + (live ? i : 0);
                    const int64_t nn = tg * 16 + sq;
                    const int64_t tile_rows = p.N - tg * 16 < 16 ? p.N - tg * 16 : 16;
                    if (c0 == 0 && coop_pre_ok) {
                        if (lane * 16 < 16 * L) {
                            uint32_t* d = reinterpret_cast<uint32_t*>(stw + lane * 16);
                            d[0] = coop_pre.w[0]; d[1] = coop_pre.w[1]; d[2] = coop_pre.w[2]; d[3] = coop_pre.w[3];
                        }
                    } else if (p.stage_stride && live) fx_stage_tile(p.ascii + tg * 16 * L, (int)tile_rows * L, stw, lane);
                    const uint8_t* crow = p.stage_stride ? stw + (nn < p.N ? sq : 0) * L : p.ascii + (nn < p.N ? nn : 0) * L;
                    float yc = 0.f;
                    fx_dense_tile8<KIND, HT, PAIR>(live, w8, lane, crow, L, p.A, p.rlh, PAIR ? 1 : 0, p.w[m] + p.off_first, nullptr,
                                                   wpair, w_d2, w_d3, db, lut_s, hx, nullptr, bad, yc);
                    if (live && w8 == 0 && g == 0 && nn < p.N) p.out[nn * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(yc);
                    if (c0 + p.coop < ncoop) __syncthreads();        // the next round reuses the exchange buffers
                }
            }
        }
    }
    fx_stamp(p.trace, 6);
    if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

template <int KIND, int A, int HT, int NT, int WAVES, bool G1, bool W1G = false, bool DG = false, bool SLAB = false, bool BT = false,
          bool PAIR = false, bool PF = false>
int launch_inst(fx_engine* e, const DenseArgs& a_in, size_t lds_bytes) {
    auto kern = k_score_dense_mfma<KIND, A, HT, NT, WAVES, G1, W1G, DG, SLAB, BT, PAIR, PF>;
    if (SLAB) lds_bytes += (size_t)2 * FX_SLAB_KG * HT * 1024;  // two slabs of KG input tiles
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    DenseArgs a = a_in;
    if (e->rows_req.on) {
        // (a launched-first call: this kernel waits for its rows tile by tile -- the lockstep slab form does not)
        if (SLAB || NT != 1) return FX_EUNSUPPORTED;
        if (e->rows_req.relay.flags) {
            // the relay: the forms that copy a tile's bytes into LDS first, one launch for the whole ensemble
            if (!((PAIR || BT) && a.stage_stride > 0) || a.M < 2 || a.m_off != 0 || a.M != a.Mtot) return FX_EUNSUPPORTED;
        }
        if (!fx_rows_plan(e)) return FX_EUNSUPPORTED;
        a.ready = e->rows_req.r;
        a.relay = e->rows_req.relay;
        e->rows_req.relay_used = a.relay.flags != nullptr;
        e->rows_req.used = true;
    }
    const int64_t U = (int64_t)a.M * a.TG;
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    const int64_t need = U;                          // small batches: one unit per workgroup (lowest latency)
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}


#if defined(FX_AB)   // (measured 11-13 % slower than the 16-wave form: lives in the A/B build only, `make ab`)
#include "score_dense_pipe_ab.h"
#endif  // FX_AB

}  // namespace

namespace {

// dense_prefetch: 1 = where it was measured to pay (a relay of at least four members), 2 = every launch that reads host rows (A/B), 0 = never
static bool fx_dense_prefetch_pays(const fx_engine* e, int M) {
    if (e->dense_prefetch == 2) return true;
    return e->dense_prefetch == 1 && e->rows_req.on && e->rows_req.relay.flags != nullptr && M >= 4;
}

// HT <= 7: 16 waves (128-register budget); HT = 8: 8 waves; HT = 13 / 16 (H <= 256): 8 waves and the HxH blocks stream from L2.
template <int HT_>
int dispatch_dense(fx_engine* e, const FxShape& s, const FxPackLayout& lay, DenseArgs& a) {
    constexpr int W = HT_ <= 7 ? 16 : 8;
    constexpr bool DGc = HT_ > 8;
    // Mid-size launches (a handful of tiles per SIMD): two waves per SIMD instead of four.  With four, a SIMD's 5-7
    // tiles go 2-2-1-1 to its waves: the first round runs four first layers at once on one LDS / VALU (7 us before the
    // first MFMA), the second round has two waves left; with two waves the tiles go 3-3 and one wave's first layer
    // overlaps the other's MFMA layers (profiles/archive/r2_trace_probe).  Long launches keep four (best steady state).
    const int64_t tiles_per_simd = (int64_t)a.M * a.TG / ((int64_t)e->num_cus * 4);
    [[maybe_unused]] const bool few_waves = e->dense_waves == 8 || (e->dense_waves == 0 && tiles_per_simd < e->dense_few_waves_below);
    const int64_t tail = DGc ? lay.off_d2 : lay.total_floats;           // end of the LDS image
    int64_t lds_from = (s.kind == FX_MLP && !e->mlp_l1_mfma) ? lay.off_w1p : 0;
    size_t lds = (size_t)(tail - lds_from) * 4 + 256 + 32;
    bool w1_global = false;
    if constexpr (!DGc) {
        if (s.kind == FX_GE && a.Lpad > 0) {
            // byte-indexed first-layer table (a.bt[] prepared by the caller) + the HxH blocks and vectors
            a.lds_from = (int)lay.off_d3;
            a.lds_floats = (int)(lay.total_floats - lay.off_d3);
            size_t need = (size_t)a.Lpad * 128 + (size_t)a.lds_floats * 4 + 256 + 32;
            const size_t stride = ((size_t)16 * s.L + 32 + 15) / 16 * 16;            // tile bytes + filler for the padded trips
#if defined(FX_AB)
            if constexpr (HT_ <= 8) {
                // software-pipelined form: one 32-position table trip per block row of the H x H layer
                const size_t pneed = (size_t)a.Lpad * 128 + (size_t)a.lds_floats * 4 + 256 + 32 + 8 * stride;
                if (e->dense_pipe && !e->rows_req.on && a.Lpad / 32 + 2 + 1 <= HT_ && s.L <= 128 && pneed <= (size_t)e->max_lds) {
                    a.stage_stride = (int)stride;
                    if (s.L <= 64) return launch_pipe<FX_GE, HT_, 1>(e, a, pneed);
                    return launch_pipe<FX_GE, HT_, 2>(e, a, pneed);
                }
            }
#endif
            if (e->stage_bytes && need + W * stride <= (size_t)e->max_lds) { a.stage_stride = (int)stride; need += W * stride; }
#if defined(FX_AB)
            if constexpr (HT_ == 7) {
                if (few_waves) return launch_inst<FX_GE, 4, HT_, 1, 8, false, false, false, false, true>(e, a, need);
            }
#endif
            if constexpr (W == 16) {
                // shared last tiles: exchange buffers behind everything else (2 x HT KiB per group of 8 waves)
                need = (need + 15) / 16 * 16;
                // (GlobalEpistasis: a lone wave's tile is cheap -- one H x H layer -- and the shared walk measured 5-7 % SLOWER
                //  (profiles/r3_dense_coop_ab.log): only dense_coop = 2 selects it, for the A/B)
                for (int groups = 2; groups >= 1 && e->dense_coop == 2 && !a.coop; --groups)
                    if (need + (size_t)groups * 2 * HT_ * 1024 <= (size_t)e->max_lds) {
                        a.coop = groups; a.coop_off = (int)(need / 4);
                        need += (size_t)groups * 2 * HT_ * 1024;
                    }
            }
            if constexpr (W == 16) {
                // rows in host memory: the next tile's bytes are asked for a tile ahead (PF: a second scratch per wave)
                // (where it pays: a relay, whose few copying waves see the link's full latency -- 8 x GE L=90: 285 -> 234 us per
                //  launch; without a relay every wave reads for itself and claiming a tile ahead costs more than it hides: MLP L=14
                //  93 -> 101 us, 3 x GE with a relay 175 -> 183: profiles/r5_dense_prefetch.log)
                if (e->ascii_host && fx_dense_prefetch_pays(e, a.M) && a.stage_stride > 0 && !a.coop && need + (size_t)W * stride <= (size_t)e->max_lds)
                    return launch_inst<FX_GE, 4, HT_, 1, W, false, false, false, false, true, false, true>(e, a, need + (size_t)W * stride);
            }
            return launch_inst<FX_GE, 4, HT_, 1, W, false, false, false, false, true>(e, a, need);
        }
    }
    if (lds > (size_t)e->max_lds) {
        // MLP with a large L*A: first-layer rows are gathered from L2 as well
        if (s.kind != FX_MLP) return FX_EUNSUPPORTED;
        lds_from = DGc ? tail : lay.off_d2;
        lds = (size_t)(tail - lds_from) * 4 + 256 + 32;
        if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
        w1_global = true;
    }
    a.lds_from = (int)lds_from;
    a.lds_floats = (int)(tail - lds_from);
    if (w1_global && fx_mlp_l1_pos_applies(e, s, lay) && (int64_t)a.M * a.TG > (int64_t)e->num_cus * e->mlp_l1_pos_tiles) {
        // (rows that arrive while the kernel runs: the caller packs first instead -- nothing has been enqueued; rows in host memory: the
        //  position-major kernel reads two to four bytes per sequence and barrier, which is no pattern for PCIe -- the planner sends such
        //  calls down the copy path, fx_score.hip, and this is the net under it)
        if (e->rows_req.on) return FX_EUNSUPPORTED;
    }
    if (w1_global && fx_mlp_l1_pos_applies(e, s, lay) && !e->ascii_host && (int64_t)a.M * a.TG > (int64_t)e->num_cus * e->mlp_l1_pos_tiles) {
        // batch launch of an MLP whose first-layer rows do not fit LDS (protein alphabets): the first layer position-major into a scratch
        // (k_mlp_l1_pos: the rows cross L2 -> LDS once per 16-32 tiles instead of once per sequence), then this kernel from there
        // in slices of at most ~512 MB of scratch (a slice of 1e5 sequences x 200 hidden units is 83 MB; the rows of a batch are independent)
        const int64_t slice_tiles = std::max<int64_t>((int64_t)e->num_cus * 16, ((int64_t)512 << 20) / ((int64_t)a.M * HT_ * 1024));
        const int64_t TG_all = a.TG, N_all = a.N;
        const uint8_t* ascii_all = a.ascii;
        float* out_all = a.out;
        void* h1 = nullptr;
        if (int rc = fx_scratch(e, 2, (size_t)a.M * (size_t)std::min(TG_all, slice_tiles) * HT_ * 64 * sizeof(f4), &h1)) return rc;
        bool first = true;
        for (int64_t t0 = 0; t0 < TG_all; t0 += slice_tiles) {
            const int64_t tiles = std::min(slice_tiles, TG_all - t0);
            a.ascii = ascii_all + t0 * 16 * a.L;
            a.out = out_all + t0 * 16 * a.out_sn;
            a.N = std::min<int64_t>(tiles * 16, N_all - t0 * 16);
            a.TG = tiles;
            L1Args l1{};
            l1.ascii = a.ascii; l1.lut = a.lut; l1.err = a.err; l1.N = a.N; l1.TG = a.TG; l1.M = a.M; l1.L = a.L; l1.A = a.A;
            l1.off_w1p = a.off_w1p; l1.off_db = a.off_db;
            for (int m = 0; m < a.M; ++m) l1.w[m] = a.w[m];
            l1.h1 = (f4*)h1;
            const int rc1 = fx_launch_mlp_l1_pos(e, l1, HT_);
            if (rc1 == FX_EUNSUPPORTED && first) break;             // (no instantiation / the slabs do not fit: the gather form below, on the whole batch)
            if (rc1 != FX_OK) return rc1;
            first = false;
            a.h1 = (const f4*)h1;
            int rc2;
            // ... and the H x H layers through LDS slabs where they do not fit (the image in LDS is empty: vectors and leftover tiles read L2)
            if (DGc && e->dense_slab && lds + (size_t)2 * FX_SLAB_KG * HT_ * 1024 <= (size_t)e->max_lds) {
                a.slab_coop = e->dense_slab_coop > 0 ? (int)(e->dense_slab_coop < 7 ? e->dense_slab_coop : 7) : 0;
                rc2 = launch_inst<FX_MLP, 4, HT_, 1, W, true, true, DGc, DGc>(e, a, lds);
            } else rc2 = launch_inst<FX_MLP, 4, HT_, 1, W, true, true, DGc>(e, a, lds);
            if (rc2 != FX_OK) return rc2;
        }
        a.ascii = ascii_all; a.out = out_all; a.N = N_all; a.TG = TG_all; a.h1 = nullptr;
        if (!first) return FX_OK;
    }
    if constexpr (DGc) {
        // hidden sizes 129..256: stream the HxH blocks through LDS slabs, one pass per round of 8 tiles (A/B: dense_slab = 0)
        // (not when the first-layer rows stream from L2 as well: measured 3 % slower there, profiles/archive/r1_run46)
        const bool slab = e->dense_slab != 0 && !e->mlp_l1_mfma && !w1_global && lds + (size_t)2 * FX_SLAB_KG * HT_ * 1024 <= (size_t)e->max_lds;
        if (slab) {
            a.slab_coop = e->dense_slab_coop > 0 ? (int)(e->dense_slab_coop < 7 ? e->dense_slab_coop : 7) : 0;
            if (s.kind == FX_MLP && fx_mlp_first_layer_form(e, s, lay) == 1) {
                // 4-letter alphabet, the pair rows fit beside the slabs (seq_len <= 15 at H = 200): a round's first layer is an LDS-bandwidth
                // gather (8 waves x seq_len rows x HT KiB), and one pre-summed row per PAIR of positions halves it
                a.lds_from = (int)tail;
                a.lds_floats = 0;
                a.off_w1pair = (int)lay.off_w1pair;
                a.pair_floats = (int)lay.pair_floats;
                return launch_inst<FX_MLP, 4, HT_, 1, W, true, false, true, true, false, true>(e, a, (size_t)lay.pair_floats * 4 + 256 + 32);
            }
            if (s.kind == FX_MLP) return launch_inst<FX_MLP, 4, HT_, 1, W, true, false, true, true>(e, a, lds);
            return launch_inst<FX_GE, 4, HT_, 1, W, false, false, true, true>(e, a, lds);
        }
    }
    if (s.kind == FX_MLP) {
        if (w1_global) return launch_inst<FX_MLP, 4, HT_, 1, W, true, true, DGc>(e, a, lds);   // gather form: A is a runtime stride
#if defined(FX_AB)
        if (e->mlp_l1_mfma) {
            if constexpr (HT_ == 7) {
                if (s.A == 4) return launch_inst<FX_MLP, 4, 7, 1, W, false, false, false>(e, a, lds);
                if (s.A == 20) return launch_inst<FX_MLP, 20, 7, 1, W, false, false, false>(e, a, lds);
            }
            return FX_EUNSUPPORTED;
        }
#endif
        if constexpr (!DGc) {
            if (e->mlp_pair && lay.off_w1pair >= 0) {
                // image = HxH blocks + vectors (the plain first-layer rows stay in global memory), then the pair rows
                const int64_t img_floats = lay.total_floats - lay.off_d2;
                size_t need = (size_t)(img_floats + lay.pair_floats) * 4 + 256 + 32;
                if (need <= (size_t)e->max_lds) {
                    const size_t stride = ((size_t)16 * s.L + 15) / 16 * 16;
                    a.lds_from = (int)lay.off_d2;
                    a.lds_floats = (int)img_floats;
                    a.off_w1pair = (int)lay.off_w1pair;
                    a.pair_floats = (int)lay.pair_floats;
#if defined(FX_AB)
                    // software-pipelined form: one pair-row gather per block row of the two H x H layers
                    if (e->dense_pipe && !e->rows_req.on && s.L / 2 + (s.L & 1) + 2 + 3 <= 2 * HT_ && s.L <= 32 && need + 8 * stride <= (size_t)e->max_lds) {
                        a.stage_stride = (int)stride;
                        return launch_pipe<FX_MLP, HT_, 1>(e, a, need + 8 * stride);
                    }
#endif
                    if (e->stage_bytes && need + W * stride <= (size_t)e->max_lds) { a.stage_stride = (int)stride; need += W * stride; }
#if defined(FX_AB)
                    if constexpr (HT_ == 7) {
                        if (few_waves) return launch_inst<FX_MLP, 4, HT_, 1, 8, true, false, false, false, false, true>(e, a, need);
                    }
#endif
                    // shared last tiles: the groups' exchange buffers (2 x HT KiB each) lie over the pair rows
                    if (W == 16 && e->dense_coop)
                        a.coop = (size_t)lay.pair_floats * 4 >= (size_t)4 * HT_ * 1024 ? 2 : (size_t)lay.pair_floats * 4 >= (size_t)2 * HT_ * 1024 ? 1 : 0;
                    if constexpr (W == 16) {
                        if (e->ascii_host && fx_dense_prefetch_pays(e, a.M) && a.stage_stride > 0 && need + (size_t)W * stride <= (size_t)e->max_lds)
                            return launch_inst<FX_MLP, 4, HT_, 1, W, true, false, false, false, false, true, true>(e, a, need + (size_t)W * stride);
                    }
                    return launch_inst<FX_MLP, 4, HT_, 1, W, true, false, false, false, false, true>(e, a, need);
                }
            }
        }
        return launch_inst<FX_MLP, 4, HT_, 1, W, true, false, DGc>(e, a, lds);
    }
    return launch_inst<FX_GE, 4, HT_, 1, W, false, false, DGc>(e, a, lds);    // A is a runtime stride for GE
}

}  // namespace

int fx_launch_score_dense_mfma(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                               float* d_out_NM, int Mtot, int m_off) {
    if (N == 0) return FX_OK;
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    for (int m = 0; m < M; ++m) {
        const FxShape& t = models[m]->shape;
        if (t.kind != s.kind || t.L != s.L || t.A != s.A || t.H != s.H) return FX_EUNSUPPORTED;
    }
    if ((s.kind != FX_MLP && s.kind != FX_GE) || M > FX_MAX_M) return FX_EUNSUPPORTED;
    if (s.A > 127) return FX_EUNSUPPORTED;                        // the gathers' bad-character test ORs the codes: needs code < 0x80
    if (!e->rows_req.on) {                                        // (a launched-first host call is big: the persistent forms only)
        const int rc = fx_launch_score_mlp_small(e, models, M, d_ascii, N, d_out_NM, Mtot, m_off);
        if (rc != FX_EUNSUPPORTED) return rc;
    }
    DenseArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    if (int rc = fx_trace_buffer(e, &a.trace)) return rc;
    a.wave_prio = (int)e->wave_prio;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.N = N; a.M = M; a.Mtot = Mtot; a.m_off = m_off; a.L = s.L; a.A = s.A; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.SG1 = lay.SG1; a.off_first = (int)lay.off_first; a.off_w1p = (int)lay.off_w1p; a.off_d2 = (int)lay.off_d2; a.off_d3 = (int)lay.off_d3;
    a.off_db = (int)lay.off_db; a.total_floats = (int)lay.total_floats;
    a.TG = (N + 15) / 16;
    if (s.kind == FX_GE && e->ge_bytetab && lay.HT <= 8) {
        int lo = 256, hi = -1;                            // the alphabet's byte range
        for (int b = 0; b < 256; ++b)
            if (e->h_lut[b] != 0xFF) { lo = b < lo ? b : lo; hi = b; }
        const int Lpad = (s.L + 31) / 32 * 32;
        const size_t need = (size_t)Lpad * 128 + (size_t)(lay.total_floats - lay.off_d3) * 4 + 256 + 32;
        if (hi >= lo && hi - lo < 32 && need <= (size_t)e->max_lds) {
            for (int m = 0; m < M; ++m) {
                fx_model* mod = models[m];
                if (!mod->d_bytetab && hipMalloc(reinterpret_cast<void**>(&mod->d_bytetab), (size_t)Lpad * 128) != hipSuccess) {
                    (void)hipGetLastError();
                    return fx_fail(e, FX_ENOMEM, "hipMalloc of the first-layer byte table failed");
                }
                if (!mod->bt_valid || std::memcmp(mod->bt_lut, e->h_lut, 256) != 0) {
                    hipLaunchKernelGGL(k_ge_bytetab, dim3((unsigned)(Lpad * 32 + 255) / 256), dim3(256), 0, e->stream,
                                       mod->d_packed + lay.off_first, e->d_lut, s.L, s.A, Lpad, lo, mod->d_bytetab);
                    FX_HIP(e, hipGetLastError());
                    std::memcpy(mod->bt_lut, e->h_lut, 256);
                    mod->bt_valid = true;
                }
                a.bt[m] = mod->d_bytetab;
            }
            a.Lpad = Lpad;
            a.bt_base = lo;
            a.validate = (m_off == 0);
        }
    }
    switch (lay.HT) {
        case 1: return dispatch_dense<1>(e, s, lay, a);
        case 2: return dispatch_dense<2>(e, s, lay, a);
        case 4: return dispatch_dense<4>(e, s, lay, a);
        case 7: return dispatch_dense<7>(e, s, lay, a);
        case 8: return dispatch_dense<8>(e, s, lay, a);
        case 13: return dispatch_dense<13>(e, s, lay, a);
        case 16: return dispatch_dense<16>(e, s, lay, a);
        default: return FX_EUNSUPPORTED;
    }
}
