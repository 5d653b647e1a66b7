// K1 (wide-alphabet form) score_cnn_pair: the fused CNN scorer for large conv3 kernels
// (protein alphabets: conv3 has A-1 = 19 taps, cnn.py:41-47).
//
// With 19 taps the register-resident sliding window of score_cnn_mfma.hip needs ~380
// registers -> one wave per SIMD.  Here TWO waves share a tile of 16 sequences and split the
// 32 filters: wave `mo` owns output-channel tile `mo` of conv2 and conv3.
//   conv1  one-hot conv as an LDS row gather (both waves, it is VALU/LDS only)
//   conv2  gather form, own output tile; the two halves of out2[t] are swapped through a
//          1 KiB LDS slot per wave (double-buffered by step parity, one barrier per step)
//   conv3  SCATTER form: out2[t] is multiplied by every tap and accumulated into a window
//          of A-1 partial output sums (own tile only: (A-1) x 4 registers); the oldest
//          slot is complete after each step and folds into the running global max.
//          Every tap hits a different accumulator -> no dependent-MFMA stalls.
// ~170 registers per wave -> 2+ waves per SIMD.  MFMA work is identical to the single-wave
// form (no recomputation); tiles are dealt round-robin so that all waves of a block take the
// same number of barriers.  Dense head: wave 0 of the pair (negligible next to the convs).
//
// SEG form (small batches: CMA-ES / DyNA-PPO populations of 1-40 protein sequences): a tile's L1 positions
// are cut into SB x PAIRS segments, one per wave pair of SB workgroups.  A pair streams its segment plus a
// halo of PL3 + PL2 positions before and PR2 + PR3 after it and pools only its own positions, so every conv3
// output sees exactly the MFMA sequence of the whole-sequence form (bit-identical results).  Segment maxima
// meet in a zeroed global pool through atomicMax on the float bits (post-ReLU values are >= 0, where float
// order == unsigned order); the last workgroup of a tile to arrive runs the dense head.  Latency of a
// 237-residue call drops from one wave walking 244 steps to ~40 steps.
#include "fx_common.h"
#include "mfma_common.h"

namespace {

struct PairArgs {
    const uint8_t* ascii;
    const uint8_t* lut;
    const float* w[FX_MAX_M];
    float* out;
    unsigned* err;
    int64_t N, TG;
    int M, Mtot, m_off;
    int64_t out_sn, out_sm;     // out[n * out_sn + column * out_sm]
    int L, rlh;
    int off_c2, off_c3, off_cb, off_w1p, conv_floats, off_d1, off_d2, off_db;
    int lds_from, lds_floats;   // LDS image = packed[lds_from .. lds_from + lds_floats): conv2, conv3, biases, conv1 rows
    int SB;                     // SEG form: workgroups per (member, tile) unit
    unsigned* pool;             // SEG form: [units][2 tiles][64 lanes][4] pooled maxima (float bits); zero between launches (fx_zero_pool with the head, memset without)
    unsigned* cnt;              // SEG form: [units] arrival counters, likewise
    // LP form (layer-parallel small batches, k_score_cnn_lp)
    f4* lp_out2;                // [unit][position][2 tiles][64 lanes]: conv2 outputs, between the two phases
    unsigned* lp_bar;           // grid barrier: counters that only ever grow, 128 bytes apart: [0] top, [1 + g] group g = block mod 16
    unsigned lp_target;         // this launch passes when the top counter reaches lp_target ...
    const FxLpMail* lp_mail;    // a PRE-LAUNCHED instance (or null): weights first, then wait for lp_word in lp_mail->req, sequences from lp_mail->bytes
    unsigned long long lp_word, lp_idle_ticks;
    unsigned* lp_state;         // pinned host word: (lp_done_seq << 1) | 1 when the instance leaves without having been asked
    unsigned* lp_decide;        // device word: the instance's ONE go / leave decision, (lp_done_seq << 2) | 1 or 2 (line 18 of the barrier counters)
    unsigned* lp_done;          // completion flag in pinned host memory (or null): the unit that finishes LAST stores lp_done_seq there
    unsigned lp_done_seq;
    unsigned lp_gtarget[16];    // ... and the LAST block of group g (the one that brings its counter to lp_gtarget[g]) arrives at the top
    int lp_nb;                  // position blocks (workgroups) per (member, tile) unit
    int lp_head_floats;         // dense head image: packed[off_d1 .. total_floats)
    int lp_rows_a;              // out2 rows of phase 2 staged behind the weights (the rest goes over the conv2 blocks)
    int lp_debug;               // profiling aid (engine option cnn_lp_debug): leave after stage k (results are then garbage)
};

// workgroups per (member, tile) unit of the position-segmented small-batch form (see launch_pair)
inline int64_t fx_pair_seg_count(int L1, int pairs, int64_t U, int64_t num_cus) {
    int64_t sb = L1 / (pairs * 2);
    if (sb > 128 / pairs) sb = 128 / pairs;
    if (sb > num_cus / U) sb = num_cus / U;
    return sb;
}

// Dense head of one tile (cnn.py:49-54): 32 pooled features -> H -> H -> 1, one wave.
template <int HT>
__device__ __forceinline__ float pair_dense_head(const f4* w_d1, const f4* w_d2, const float* db, f4 pool0, f4 pool1,
                                                 int lane, int g, int rlh) {
    asm volatile("" : "+v"(w_d1), "+v"(w_d2), "+v"(db));   // keep block addresses out of the tile loop's live set
    f4 pooled[2][1];
    pooled[0][0] = pool0;
    pooled[1][0] = pool1;
    f4 h1[HT][1], h2[HT][1];
    init_bias<HT, 1>(db, h1, g);
    mma_layer<2, HT, 1>(w_d1, pooled, h1, lane);
    relu_tiles<HT, 1>(h1);
    init_bias<HT, 1>(db + 16 * HT, h2, g);
    mma_layer<HT, HT, 1>(w_d2, h1, h2, lane, rlh);
    relu_tiles<HT, 1>(h2);
    float y[1];
    final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2, y, g);
    return y[0];
}

// HEAD = false: the conv part only -- each wave leaves its half of the tile's pooled features in `pool`
// ([unit][2 tiles][64 lanes] f4, the layout of score_cnn_split.hip; with SEG the segments meet there through atomicMax on
// the float bits of a zeroed pool) and k_cnn_head finishes the sequence.
template <int A, int K, int HT, int WAVES, bool SEG, bool HEAD = true>
__global__ void __launch_bounds__(WAVES * 64) k_score_cnn_pair(PairArgs p) {
    constexpr int FT = 2, K3 = A - 1, PAIRS = WAVES / 2;
    constexpr int PL2 = (K - 1) / 2, PR2 = K - 1 - PL2;
    constexpr int PL3 = (K3 - 1) / 2, PR3 = K3 - 1 - PL3;
    constexpr int TAPG = 4;                               // conv3 taps whose weights are in flight together
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave >> 1, mo = wave & 1;
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L, L1 = L - K + 1;
    f4* xbuf = reinterpret_cast<f4*>(smem + p.lds_floats);             // [3: parity 0, parity 1, pooled][PAIRS][2 tiles][64 lanes]
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem + p.lds_floats + 3 * PAIRS * 2 * 64 * 4);

    for (int i = tid; i < 64; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];

    const int sb = SEG ? (int)(blockIdx.x % p.SB) : 0;
    int64_t u_lo, u_hi;
    if (SEG) { u_lo = (int64_t)(blockIdx.x / p.SB); u_hi = u_lo + 1; }
    else fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    bool bad = false;

    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();
        {
            const f4* src = reinterpret_cast<const f4*>(p.w[m] + p.lds_from);
            f4* dst = reinterpret_cast<f4*>(smem);
            fill_lds(dst, src, p.lds_floats / 4);
        }
        __syncthreads();
        const f4* w_c2 = reinterpret_cast<const f4*>(smem + (p.off_c2 - p.lds_from));
        const f4* w_c3 = reinterpret_cast<const f4*>(smem + (p.off_c3 - p.lds_from));
        const float* cb = smem + (p.off_cb - p.lds_from);
        const float* w1p = smem + (p.off_w1p - p.lds_from);
        const f4* w_d1 = reinterpret_cast<const f4*>(p.w[m] + p.off_d1);   // dense head streams from L2
        const f4* w_d2 = reinterpret_cast<const f4*>(p.w[m] + p.off_d2);
        const float* db = p.w[m] + p.off_db;

        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        const int iters = SEG ? 1 : (int)((t_hi - t_lo + PAIRS - 1) / PAIRS);

        for (int it = 0; it < iters; ++it) {
            const int64_t tg = SEG ? t_lo : t_lo + (int64_t)it * PAIRS + pair;   // SEG: all pairs share the tile
            const bool live = tg < t_hi;                 // idle pairs run along (barriers) on sequence 0
            const int64_t n = tg * 16 + sq;
            const uint8_t* row = p.ascii + ((live && n < p.N) ? n : 0) * L;

            const int steps = L1 + PR2 + PR3;
            // positions this pair pools, the step it starts at, and the (block-uniform) number of steps
            int seg_lo = 0, seg_hi = L1, s0 = 0, s_end = steps, nsteps = steps;
            if (SEG) {
                const int S = p.SB * PAIRS, q = sb * PAIRS + pair;
                seg_lo = __builtin_amdgcn_readfirstlane((int)((int64_t)L1 * q / S));       // (wave-uniform: scalar tap tests below)
                seg_hi = __builtin_amdgcn_readfirstlane((int)((int64_t)L1 * (q + 1) / S));
                s0 = seg_lo - PL3 - PL2 > 0 ? seg_lo - PL3 - PL2 : 0;
                s_end = seg_hi + PR2 + PR3 < steps ? seg_hi + PR2 + PR3 : steps;
                nsteps = (L1 + S - 1) / S + PL3 + PL2 + PR2 + PR3;
            }
            const int c2_from = seg_lo - PL3 > 0 ? seg_lo - PL3 : 0;   // first conv2 position this pair needs

            int cw[K];
#pragma unroll
            for (int j = 0; j < K - 1; ++j) {
                int c = lut_s[row[s0 + j]];
                if (c == 0xFF) { bad |= live; c = 0; }
                cw[j + 1] = c;
            }
            f4 win1[K][FT], accw[K3], gmax = splat4(0.f);
            const f4 bias3 = *reinterpret_cast<const f4*>(&cb[32 * FT + 16 * mo + 4 * g]);
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int t = 0; t < FT; ++t) win1[j][t] = splat4(0.f);
#pragma unroll
            for (int j = 0; j < K3; ++j) accw[j] = bias3;

            for (int k = 0; k < nsteps; ++k) {
                const int s = s0 + k;
                const bool on = !SEG || s < s_end;       // SEG: pairs with a shorter range idle along (barriers)
                asm volatile("" ::: "memory");           // keep the LDS weight reads inside the position loop
#pragma unroll
                for (int j = 0; j < K - 1; ++j) {
                    cw[j] = cw[j + 1];
#pragma unroll
                    for (int t = 0; t < FT; ++t) win1[j][t] = win1[j + 1][t];
                }
                // ---- conv1 (valid) at t1 = s: gather of K kernel rows, both channel tiles
                if (on && s < L1) {
                    int c = lut_s[row[s + K - 1]];
                    if (c == 0xFF) { bad |= live; c = 0; }
                    cw[K - 1] = c;
                    f4 o1[FT];
#pragma unroll
                    for (int t = 0; t < FT; ++t) o1[t] = *reinterpret_cast<const f4*>(&cb[16 * t + 4 * g]);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const float* rowp = w1p + (j * A + cw[j]) * FX_C1_ROW(FT) + 4 * g;
#pragma unroll
                        for (int t = 0; t < FT; ++t) o1[t] += *reinterpret_cast<const f4*>(rowp + 16 * t);
                    }
#pragma unroll
                    for (int t = 0; t < FT; ++t) win1[K - 1][t] = relu4(o1[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < FT; ++t) win1[K - 1][t] = splat4(0.f);
                }

                // ---- conv2 (same) at t2 = s - PR2, own output tile; two partial chains (one per input tile)
                const int t2 = s - PR2;
                const bool c2 = on && t2 >= c2_from && t2 < L1;
                f4 mine = splat4(0.f);
                f4* slot = xbuf + (((k & 1) * PAIRS + pair) * 2) * 64;
                if (c2) {
                    f4 o2a = *reinterpret_cast<const f4*>(&cb[16 * FT + 16 * mo + 4 * g]);
                    f4 o2b = splat4(0.f);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        // out-of-range taps read the zeros the window holds there
                        const f4 a0 = w_c2[((j * FT + 0) * FT + mo) * 64 + lane];
                        const f4 a1 = w_c2[((j * FT + 1) * FT + mo) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o2a = mfma16(a0[r], win1[j][0][r], o2a);
                            o2b = mfma16(a1[r], win1[j][1][r], o2b);
                        }
                    }
                    mine = relu4(o2a + o2b);
                    // ---- swap halves with the partner wave (slot per parity: one barrier per step)
                    slot[mo * 64 + lane] = mine;
                    if (!SEG) __syncthreads();            // whole-sequence form: c2 is block-uniform
                }
                if (SEG) __syncthreads();                 // segment form: ranges differ per pair -> barrier every step
                if (c2) {
                    const f4 theirs = slot[(1 - mo) * 64 + lane];
                    f4 out2[FT];                          // (no runtime-indexed register arrays: they go to scratch)
                    out2[0] = mo == 0 ? mine : theirs;
                    out2[1] = mo == 0 ? theirs : mine;

                    // ---- conv3 (same, A-1 taps), scatter form: tap j feeds output position t2 - j + PL3,
                    //      which lives in window slot K3-1-j (slot i <-> position t2 - PR3 + i)
                    if constexpr (SEG) {
                        // Segment form: only the taps that land on this pair's OWN positions are issued -- a halo step
                        // feeds at most (segment length) of the A-1 window slots that will ever be pooled; the other
                        // slots may hold anything.  Every own output still receives all its taps, in the same order:
                        // same bits as the whole-sequence walk.  (A segment of 2-4 positions has 22 halo steps at
                        // A = 20: conv3 drops from 152 to 16-32 MFMAs per wave per step.)
                        const int j_lo = t2 + PL3 - seg_hi + 1, j_hi = t2 + PL3 - seg_lo;
#pragma unroll
                        for (int j = 0; j < K3; ++j) {
                            if (j >= j_lo && j <= j_hi) {
                                asm volatile("" ::: "memory");
                                f4 a[FT];
#pragma unroll
                                for (int mi = 0; mi < FT; ++mi) a[mi] = w_c3[((j * FT + mi) * FT + mo) * 64 + lane];
#pragma unroll
                                for (int mi = 0; mi < FT; ++mi)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) accw[K3 - 1 - j] = mfma16(a[mi][r], out2[mi][r], accw[K3 - 1 - j]);
                            }
                        }
                    } else
#pragma unroll
                    for (int j0 = 0; j0 < K3; j0 += TAPG) {
                        // fence the weight reads of each tap group: without it the scheduler hoists all
                        // 2*(A-1) blocks ahead of the MFMAs and spills ~100 registers
                        asm volatile("" ::: "memory");
                        f4 a[TAPG][FT];
#pragma unroll
                        for (int jj = 0; jj < TAPG; ++jj)
#pragma unroll
                            for (int mi = 0; mi < FT; ++mi)
                                if (j0 + jj < K3) a[jj][mi] = w_c3[(((j0 + jj) * FT + mi) * FT + mo) * 64 + lane];
#pragma unroll
                        for (int mi = 0; mi < FT; ++mi)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int jj = 0; jj < TAPG; ++jj) {
                                    const int j = j0 + jj;
                                    // (slots whose position falls outside [0, L1) are simply never pooled:
                                    //  no per-tap branch -> straight-line MFMA stream)
                                    if (j < K3) accw[K3 - 1 - j] = mfma16(a[jj][mi][r], out2[mi][r], accw[K3 - 1 - j]);
                                }
                    }
                }
                // ---- slot 0 (position t2 - PR3) is complete: GlobalMaxPooling1D of relu(conv3), then slide
                const int t3f = t2 - PR3;
                if (t3f >= seg_lo && t3f < seg_hi) gmax = pool_max4(gmax, accw[0]);
#pragma unroll
                for (int j = 0; j < K3 - 1; ++j) accw[j] = accw[j + 1];
                accw[K3 - 1] = bias3;
            }

            if constexpr (!HEAD && !SEG) {
                if (live) reinterpret_cast<f4*>(p.pool)[(((int64_t)m * p.TG + tg) * 2 + mo) * 64 + lane] = gmax;
            } else if constexpr (!HEAD && SEG) {
                unsigned* pl = p.pool + ((((int64_t)m * p.TG + tg) * 2 + mo) * 64 + lane) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicMax(&pl[r], __float_as_uint(gmax[r]));
            } else if (!SEG) {
                // ---- pooled features: swap halves once more, then wave 0 of the pair runs the dense head
                f4* pslot = xbuf + ((2 * PAIRS + pair) * 2) * 64;  // dedicated slot: no reuse hazard with the step slots
                pslot[mo * 64 + lane] = gmax;
                __syncthreads();
                if (mo == 0) {                            // (pslot is rewritten only after the next tile's ~L barriers)
                    const float y = pair_dense_head<HT>(w_d1, w_d2, db, gmax, pslot[64 + lane], lane, g, p.rlh);
                    if (g == 0 && live && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y);
                }
            } else {
                // ---- segment maxima meet in the global pool; the last workgroup of the tile runs the head
                const int64_t unit = (int64_t)m * p.TG + tg;
                unsigned* pl = p.pool + ((unit * 2 + mo) * 64 + lane) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicMax(&pl[r], __float_as_uint(gmax[r]));
                __syncthreads();                                          // (every wave's atomicMax has been performed: vmcnt(0))
                int* last = reinterpret_cast<int*>(xbuf);                 // step slots are idle now
                if (tid == 0) {
                    __threadfence();                                      // ONE fence per workgroup (an L2 write-back each, serialised per XCD)
                    *last = (atomicAdd(&p.cnt[unit], 1u) == (unsigned)p.SB - 1u) ? 1 : 0;
                }
                __syncthreads();
                if (*last && wave == 0) {
                    __threadfence();
                    const unsigned* p0 = p.pool + ((unit * 2 + 0) * 64 + lane) * 4;
                    f4 pool0, pool1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                         // device-coherent reads; the entries go back to zero
                        pool0[r] = __uint_as_float(__hip_atomic_load(&p0[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        pool1[r] = __uint_as_float(__hip_atomic_load(&p0[256 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        __hip_atomic_store(const_cast<unsigned*>(&p0[r]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(const_cast<unsigned*>(&p0[256 + r]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) __hip_atomic_store(&p.cnt[unit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float y = pair_dense_head<HT>(w_d1, w_d2, db, pool0, pool1, lane, g, p.rlh);
                    if (g == 0 && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y);
                }
            }
        }
    }
    if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

#include "score_cnn_lp.h"   // k_score_cnn_lp + launch_lp (the layer-parallel small-batch form)

template <int A, int K, int HT, int WAVES>
int launch_pair(fx_engine* e, PairArgs a, size_t lds_bytes) {
    auto whole = k_score_cnn_pair<A, K, HT, WAVES, false>;
    auto seg = k_score_cnn_pair<A, K, HT, WAVES, true>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(whole), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(seg), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    const int64_t U = (int64_t)a.M * a.TG;
    const int L1 = a.L - K + 1;
    if (e->cnn_pair_seg < 0 && !e->rows_req.on && U > e->num_cus && U <= 2 * (int64_t)e->num_cus && a.TG >= 2 && a.out) {
        // Between one and two units per CU (round 6) the whole-sequence form below is ONE serial walk per wave pair with half the pairs idle
        // (1.47 ms at 237 residues whatever the count); two launches of the segmented form, half the tiles each, are two walks of a third
        // of the length (3 x 2000 GFP sequences: 1466 -> 960 us)
        PairArgs lo = a, hi = a;
        const int64_t t_half = (a.TG + 1) / 2;
        lo.TG = t_half; lo.N = a.N < t_half * 16 ? a.N : t_half * 16;
        hi.TG = a.TG - t_half; hi.N = a.N - t_half * 16; hi.ascii = a.ascii + t_half * 16 * a.L; hi.out = a.out + t_half * 16 * a.out_sn;
        if (int rc = launch_pair<A, K, HT, WAVES>(e, lo, lds_bytes)) return rc;
        return launch_pair<A, K, HT, WAVES>(e, hi, lds_bytes);
    }
    // Small batch (a CMA-ES / DyNA-PPO population, a single sequence): fewer units than half the CUs.  Cut every
    // tile into segments over SB workgroups so that the call's latency is ~L1/S + halo steps.  The halo (PL3 + PL2 +
    // PR2 + PR3 = 22 positions at A = 20) is recomputed by every segment, but the machine is otherwise empty: as many
    // workgroups per tile as fit in ONE wave of the grid (U x SB <= CUs), down to segments of two positions
    // (tools/archive/runs/r2_pair_seg_sweep.py: a 1-16 sequence call at L = 237 250 -> 181 us, at L = 90 269 -> 169 us).
    int64_t sb = fx_pair_seg_count(L1, WAVES / 2, U, e->num_cus);
    // (round 6: up to ONE unit per CU -- between half and all of the CUs a workgroup's four wave pairs still split its tile's positions, SB = 1:
    //  3 x 1000 GFP sequences were 189 lone 1.4 ms walks by one pair per workgroup)
    if (e->cnn_pair_seg == 0 || U > e->num_cus) sb = 0;
    if (e->cnn_pair_seg > 0) sb = e->cnn_pair_seg;                       // test knob: force SB
    if (sb >= 1) {
        if constexpr (A == 20 && K == 5 && HT == 7 && WAVES == 8) {
            // layer-parallel form first (no halo recomputation): canonical protein CNN, units x position blocks <= CUs
            if (e->cnn_pair_seg < 0) {
                const int rc_lp = launch_lp<A, K, HT>(e, a, lds_bytes);
                if (rc_lp != FX_EUNSUPPORTED) return rc_lp;
            }
        }
        void* ws = nullptr;
        const size_t pool_bytes = (size_t)U * 2 * 64 * 4 * sizeof(unsigned), cnt_bytes = (size_t)U * sizeof(unsigned);
        int rc = fx_zero_pool(e, pool_bytes + cnt_bytes, &ws);   // all zeros between launches: the head workgroup resets what it read
        if (rc) return rc;
        a.pool = (unsigned*)ws;
        a.cnt = (unsigned*)((char*)ws + pool_bytes);
        if constexpr (A == 20 && K == 5 && HT == 7 && WAVES == 8) {
            // Same number of segments from twice the workgroups of half the size, when they still fit in one wave of the
            // grid: ONE wave per SIMD instead of two.  A step of a wave is 192 MFMAs = 2.6 us of its SIMD's pipe, two
            // waves on a SIMD take turns, and a call of a few sequences is a chain of ~25 such steps.
            const int64_t sb4 = fx_pair_seg_count(L1, 2, U, e->num_cus);
            if (e->cnn_pair_seg < 0 && e->cnn_pair_seg4 && sb4 * 2 >= sb * 4) {
                auto seg4 = k_score_cnn_pair<A, K, HT, 4, true>;
                static bool attr4[64] = {};
                if (!attr4[e->device & 63]) {
                    FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(seg4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr4[e->device & 63] = true;
                }
                a.SB = (int)sb4;
                hipLaunchKernelGGL(seg4, dim3((unsigned)(U * sb4)), dim3(4 * 64), lds_bytes, e->stream, a);
                FX_HIP(e, hipGetLastError());
                return FX_OK;
            }
        }
        a.SB = (int)sb;
        hipLaunchKernelGGL(seg, dim3((unsigned)(U * sb)), dim3(WAVES * 64), lds_bytes, e->stream, a);
        FX_HIP(e, hipGetLastError());
        return FX_OK;
    }
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    const int64_t need = U;                          // one tile per workgroup at most
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(whole, dim3((unsigned)blocks), dim3(WAVES * 64), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

template <int K>
int launch_pair_conv(fx_engine* e, PairArgs a, size_t lds_bytes) {
    constexpr int WAVES = 8;
    auto kern = k_score_cnn_pair<20, K, 1, WAVES, false, false>;
    auto seg = k_score_cnn_pair<20, K, 1, WAVES, true, false>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(seg), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    const int64_t U = (int64_t)a.M * a.TG;
    // small batch: position-segmented over SB workgroups per tile, as launch_pair does for the fused form
    const int L1 = a.L - K + 1;
    int64_t sb = fx_pair_seg_count(L1, WAVES / 2, U, e->num_cus);
    if (e->cnn_pair_seg == 0 || U > e->num_cus) sb = 0;
    if (e->cnn_pair_seg > 0) sb = e->cnn_pair_seg;
    if (sb >= 1) {
        FX_HIP(e, hipMemsetAsync(a.pool, 0, (size_t)U * 2 * 64 * 4 * sizeof(unsigned), e->stream));   // maxima of relu outputs: >= +0
        a.SB = (int)sb;
        hipLaunchKernelGGL(seg, dim3((unsigned)(U * sb)), dim3(WAVES * 64), lds_bytes, e->stream, a);
        FX_HIP(e, hipGetLastError());
        return FX_OK;
    }
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    if (blocks > U) blocks = U;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

}  // namespace

// Conv part only of a 20-letter CNN (kernel_size 2..7, two channel tiles): pooled features to `d_pool`
// ([member * TG + tile][2][64 lanes] f4); score_cnn_split.hip runs the head kernel on them.
int fx_launch_cnn_pair_conv(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, void* d_pool) {
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    if (lay.FT != 2 || s.A != 20 || M > FX_MAX_M || s.K < 2 || s.K > 7) return FX_EUNSUPPORTED;
    constexpr int WAVES = 8;
    const size_t lds = (size_t)(lay.conv_floats - lay.off_c2) * 4 + (size_t)3 * (WAVES / 2) * 2 * 64 * 16 + 256 + 16;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    PairArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = nullptr; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.N = N; a.TG = (N + 15) / 16; a.M = M; a.Mtot = M; a.m_off = 0; a.L = s.L; a.rlh = 4;
    a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3; a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p;
    a.conv_floats = (int)lay.conv_floats; a.lds_from = (int)lay.off_c2; a.lds_floats = (int)(lay.conv_floats - lay.off_c2);
    a.off_d1 = (int)lay.off_d1; a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db;
    a.pool = (unsigned*)d_pool;
    switch (s.K) {
        case 2: return launch_pair_conv<2>(e, a, lds);
        case 3: return launch_pair_conv<3>(e, a, lds);
        case 4: return launch_pair_conv<4>(e, a, lds);
        case 5: return launch_pair_conv<5>(e, a, lds);
        case 6: return launch_pair_conv<6>(e, a, lds);
        case 7: return launch_pair_conv<7>(e, a, lds);
        default: return FX_EUNSUPPORTED;
    }
}

int fx_launch_score_cnn_pair(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                             float* d_out_NM, int Mtot, int m_off) {
    if (N == 0) return FX_OK;
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    if (lay.FT != 2 || s.A != 20 || M > FX_MAX_M) return FX_EUNSUPPORTED;
    if (s.K != 5 && !((s.K == 3 || s.K == 7) && lay.HT == 7)) return FX_EUNSUPPORTED;
    constexpr int WAVES = 8;
    const size_t lds = (size_t)(lay.conv_floats - lay.off_c2) * 4 + (size_t)3 * (WAVES / 2) * 2 * 64 * 16 + 256 + 16;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    PairArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.N = N; a.TG = (N + 15) / 16; a.M = M; a.Mtot = Mtot; a.m_off = m_off; a.L = s.L; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3; a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p;
    a.conv_floats = (int)lay.conv_floats; a.lds_from = (int)lay.off_c2; a.lds_floats = (int)(lay.conv_floats - lay.off_c2); a.off_d1 = (int)lay.off_d1; a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db;
    a.lp_head_floats = (lay.off_d1 % 4 == 0) ? (int)(lay.total_floats - lay.off_d1) : 0;   // (0: no layer-parallel form)
    if (s.K == 3) return launch_pair<20, 3, 7, WAVES>(e, a, lds);
    if (s.K == 7) return launch_pair<20, 7, 7, WAVES>(e, a, lds);
    switch (lay.HT) {
        case 1: return launch_pair<20, 5, 1, WAVES>(e, a, lds);
        case 2: return launch_pair<20, 5, 2, WAVES>(e, a, lds);
        case 4: return launch_pair<20, 5, 4, WAVES>(e, a, lds);
        case 7: return launch_pair<20, 5, 7, WAVES>(e, a, lds);
        case 8: return launch_pair<20, 5, 8, WAVES>(e, a, lds);
        case 13: return launch_pair<20, 5, 13, WAVES>(e, a, lds);
        case 16: return launch_pair<20, 5, 16, WAVES>(e, a, lds);
        default: return FX_EUNSUPPORTED;
    }
}
