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
                        const float* rowp = w1p + (j * A + cw[j]) * (16 * FT) + 4 * g;
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

#define LP_POOLS 16            // sub-pools per unit of the layer-parallel form's max-pool meeting point

// LP form (round 4): small protein batches LAYER BY LAYER across the chip instead of position segments with halos.
//
// A CMA-ES population / DyNA-PPO environment batch is 1-40 sequences of 90-237 residues: 1-3 tiles per member.  The SEG form
// above cuts a tile's positions into segments of two and pays for it with the halo: conv3 reaches 9 + 9 positions and conv2
// 2 + 2 around every output, so a pair walks 24 steps (barrier + LDS exchange + ~70 MFMAs of a lone wave each, ~2.5 us) for
// its 2 positions -- 73 us for one 237-residue tile whose whole arithmetic is ~90 k MFMAs = 1.2 us of the machine
// (profiles/archive/r2_protein_small_calls.log).  Here nothing is recomputed:
//   phase 1  workgroup (unit, block b) computes conv1 + conv2 for ITS positions (conv1, a row gather, also for the 2 + 2
//            neighbours conv2 reaches) and leaves out2[position] in device memory (2 KiB per position and tile);
//   barrier  all workgroups of the launch (a counter in device memory, agent-scope release / acquire: one per launch);
//   phase 2  the same workgroup computes conv3 for its positions from out2[position - 9 .. position + 9] (L2 reads),
//            pools them, and the blocks of a unit meet in the zeroed pool (atomicMax on the float bits, ticket) as the SEG
//            form's workgroups do; the last one runs the dense head.
// A wave owns one output-channel tile (mo) and up to PBW positions; per output element the MFMA sequence is the pair
// kernel's: conv2 = bias + taps 0..K-1 x (k-steps 0..3) as two chains (input tiles 0 / 1) added at the end, conv3 = bias +
// the conv2 outputs in position order (= tap order) x (input tile, k-step) -- out-of-range positions are skipped for conv3 and
// read as zeros for conv2, exactly as there -- so the scores are the SAME BITS as k_score_cnn_pair's (tested).
// Needs every workgroup co-resident (grid <= CUs, ~111 KiB of LDS each: the launcher checks); a barrier not passed within
// 1 s raises FX_ERR_TIMEOUT instead of hanging the device.
template <int A, int K, int HT>
__global__ void __launch_bounds__(256) k_score_cnn_lp(PairArgs p) {
    constexpr int FT = 2, K3 = A - 1, PBW = 8;
    constexpr int PL2 = (K - 1) / 2, PR2 = K - 1 - PL2;
    constexpr int PL3 = (K3 - 1) / 2, PR3 = K3 - 1 - PL3;
    static_assert(LP_POOLS == 16, "the head folds 2 x 8 sub-pools");
    constexpr int SPAN = PBW + K - 1 + K - 1;               // sequence bytes a wave's phase 1 reads per sequence (<= 16)
    static_assert(SPAN <= 16, "the byte rows of phase 1 are staged 16 per sequence");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mo = wave & 1, half = wave >> 1;
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L, L1 = L - K + 1, NB = p.lp_nb;
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem + p.lds_floats);
    int* flags = reinterpret_cast<int*>(lut_s + 256);
    uint8_t* rows_s = lut_s + 256 + 16 + wave * 256;        // this wave's 16 sequences x 16 bytes
    for (int i = tid; i < 64; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];
    const int64_t unit = blockIdx.x / (unsigned)NB;
    const int b = (int)(blockIdx.x % (unsigned)NB);
    const int m = (int)(unit / p.TG);
    const int64_t tg = unit - (int64_t)m * p.TG;
    // this block's positions, cut in two for the wave pairs (half 0 / 1); a wave owns output tile `mo` of them
    const int P0 = (int)((int64_t)L1 * b / NB), P1 = (int)((int64_t)L1 * (b + 1) / NB);
    const int h0 = __builtin_amdgcn_readfirstlane(P0 + (P1 - P0) * half / 2), h1 = __builtin_amdgcn_readfirstlane(P0 + (P1 - P0) * (half + 1) / 2);
    const int64_t n = tg * 16 + sq;
    const bool live = n < p.N;
    const uint8_t* row = p.ascii + (live ? n : 0) * L;
    const int s0 = h0 - PL2 > 0 ? h0 - PL2 : 0;
    // the sequence bytes of phase 1, all requested at once (a byte per step from global memory is a ~1.5 us round trip per step):
    // lane (sq, g) brings bytes 4g .. 4g+3 of its sequence's span [s0, s0 + 16)
    auto read_rows = [&]() {
        uint8_t rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int at = s0 + 4 * g + i; rb[i] = row[at < L ? at : L - 1]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) rows_s[sq * 16 + 4 * g + i] = rb[i];
    };
    if (!p.lp_mail) read_rows();                             // (a pre-launched instance has no sequences yet)
    // Two-part LDS fill: what phase 1 reads (conv2 blocks; biases + conv1 rows behind conv3) now, conv3 (78 of the 111 KiB) after
    // phase 1 -- its loads are in flight while the block waits at the barrier anyway
    const int a_lo = 0, a_hi = p.off_c3 - p.lds_from, c_lo = p.off_cb - p.lds_from, c_hi = p.lds_floats;
    fill_lds(reinterpret_cast<f4*>(smem + a_lo), reinterpret_cast<const f4*>(p.w[m] + p.lds_from + a_lo), (a_hi - a_lo) / 4);
    fill_lds(reinterpret_cast<f4*>(smem + c_lo), reinterpret_cast<const f4*>(p.w[m] + p.lds_from + c_lo), (c_hi - c_lo) / 4);
    __syncthreads();
    if (p.lp_mail) {
        // PRE-LAUNCHED instance: the weights are in LDS; wait for this instance's request word (the host stores it through the BAR
        // once the caller is back with its sequences), or leave: told to (another call shape, another kernel wants the CUs), or
        // nobody came within the idle window.  Leaving touches neither the barrier counters nor the pools: the host puts the
        // counters back when it finds the instance gone.
        if (tid == 0) {
            const unsigned long long* w = &p.lp_mail->req[blockIdx.x & 15u].w;
            const unsigned long long t0 = wall_clock64();
            // ONE decision for the whole instance (round-4 advisor finding): every block used to decide on its own clock and its own
            // copy of the request word, so a host thread descheduled between the 16 word stores -- or a post landing as the idle window
            // closed -- could leave some blocks gone and others waiting at the grid barrier for them (FX_ERR_TIMEOUT after 1 s).  The
            // first block to see a reason to go or to leave publishes it in device memory -- (sequence number << 2) | 1 go / 2 leave,
            // compare-and-swap from whatever an older instance left there -- and every block, that one included, does what the word says.
            const unsigned tag = p.lp_done_seq << 2;
            int go = 0;
            for (;;) {
                unsigned d = __hip_atomic_load(p.lp_decide, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((d & ~3u) == tag && (d & 3u)) { go = (d & 3u) == 1u; break; }
                const unsigned long long r = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                unsigned mine = 0;
                if (r == p.lp_word) mine = tag | 1u;
                else if (r == (p.lp_word | 0xFFFFull) || wall_clock64() - t0 > p.lp_idle_ticks) mine = tag | 2u;      // (told to leave: ITS sequence number with 0xFFFF sequences)
                if (mine) { (void)__hip_atomic_compare_exchange_strong(p.lp_decide, &d, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
                __builtin_amdgcn_s_sleep(2);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            flags[2] = go;
            if (!go && blockIdx.x == 0) __hip_atomic_store(p.lp_state, (p.lp_done_seq << 1) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (!flags[2]) return;
        read_rows();                                         // (first touch of these lines by this launch: written by the host just now)
    }
    if (p.lp_debug == 1) return;
    const f4* w_c2 = reinterpret_cast<const f4*>(smem + (p.off_c2 - p.lds_from));
    const f4* w_c3 = reinterpret_cast<const f4*>(smem + (p.off_c3 - p.lds_from));
    const float* cb = smem + (p.off_cb - p.lds_from);
    const float* w1p = smem + (p.off_w1p - p.lds_from);
    f4* o2g = p.lp_out2 + (unit * L1) * 2 * 64;            // [position][tile][lane]
    bool bad = false;

    // ---- phase 1: conv1 (valid, row gather) over [h0 - PL2, h1 + PR2), conv2 (same) at [h0, h1): the pair kernel's step loop
    if (h1 > h0) {
        const int s_last = h1 - 1 + PR2;
        const uint8_t* rs = rows_s + sq * 16 - s0;           // rs[position] = the sequence's byte there (LDS)
        int cw[K];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {
            int c = lut_s[rs[s0 + j]];
            if (c == 0xFF) { bad |= live; c = 0; }
            cw[j + 1] = c;
        }
        f4 win1[K][FT];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int t = 0; t < FT; ++t) win1[j][t] = splat4(0.f);
        for (int s = s0; s <= s_last; ++s) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < K - 1; ++j) {
                cw[j] = cw[j + 1];
#pragma unroll
                for (int t = 0; t < FT; ++t) win1[j][t] = win1[j + 1][t];
            }
            if (s < L1) {
                int c = lut_s[rs[s + K - 1]];
                if (c == 0xFF) { bad |= live; c = 0; }
                cw[K - 1] = c;
                f4 o1[FT];
#pragma unroll
                for (int t = 0; t < FT; ++t) o1[t] = *reinterpret_cast<const f4*>(&cb[16 * t + 4 * g]);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float* rowp = w1p + (j * A + cw[j]) * (16 * FT) + 4 * g;
#pragma unroll
                    for (int t = 0; t < FT; ++t) o1[t] += *reinterpret_cast<const f4*>(rowp + 16 * t);
                }
#pragma unroll
                for (int t = 0; t < FT; ++t) win1[K - 1][t] = relu4(o1[t]);
            } else {
#pragma unroll
                for (int t = 0; t < FT; ++t) win1[K - 1][t] = splat4(0.f);
            }
            const int t2 = s - PR2;
            if (t2 >= h0 && t2 < L1) {                       // (t2 < h1 by the loop bound)
                f4 o2a = *reinterpret_cast<const f4*>(&cb[16 * FT + 16 * mo + 4 * g]);
                f4 o2b = splat4(0.f);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const f4 a0 = w_c2[((j * FT + 0) * FT + mo) * 64 + lane];
                    const f4 a1 = w_c2[((j * FT + 1) * FT + mo) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o2a = mfma16(a0[r], win1[j][0][r], o2a);
                        o2b = mfma16(a1[r], win1[j][1][r], o2b);
                    }
                }
                fx_store16_agent(&o2g[(t2 * 2 + mo) * 64 + lane], relu4(o2a + o2b));   // (written through: read by workgroups on other XCDs)
            }
        }
    }
    if (p.lp_debug == 2) return;
    // the conv3 blocks: requested now, they land while the block waits for the others
    fill_lds(reinterpret_cast<f4*>(smem + a_hi), reinterpret_cast<const f4*>(p.w[m] + p.lds_from + a_hi), (c_lo - a_hi) / 4);

    // ---- barrier over the whole launch.  The conv2 outputs were written through (sc1 stores) and will be read past the
    //      non-coherent cache levels (sc1 loads), so the hand-off needs no L2 write-back / invalidate -- with a release and an
    //      acquire fence per workgroup the 243 workgroups of a 40-sequence call spent 21 us here (a fence is ~0.5 us and the
    //      fences of an XCD serialise, profiles/r4_lp_stages.log): every wave waits for its own stores, then one counter.
    fx_wait_vm(0);
    __syncthreads();
    if (p.lp_debug == 3) return;
    if (tid == 0) {
        // two levels: 243 arrivals at ONE counter serialise at the memory side (~40-85 ns per same-address atomic: 10 us);
        // a block arrives at its group's counter (16 groups, a line each), the last of a group at the top one
        const int grp = (int)(blockIdx.x & 15u);
        const unsigned before = __hip_atomic_fetch_add(p.lp_bar + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == p.lp_gtarget[grp]) __hip_atomic_fetch_add(p.lp_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        int timed_out = 0;
        while ((int)(__hip_atomic_load(p.lp_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.lp_target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000ull) { timed_out = 1; break; }    // 1 s at 100 MHz
        }
        flags[0] = timed_out;
    }
    __syncthreads();
    if (flags[0]) {
        if (tid == 0) fx_raise(p.err, FX_ERR_TIMEOUT);
        return;
    }
    if (p.lp_debug == 4) return;
    // ---- phase 2: conv3 (same, A - 1 taps) at the wave's positions from out2[position - PL3 .. position + PR3], pooled.
    //      The rows the BLOCK needs -- [P0 - PL3, P1 + PR3), 2 KiB each -- are staged into LDS by all four waves at once (every
    //      load in flight together: one L2 round trip instead of one per row), into the space behind the weights and over the
    //      conv2 blocks, which nobody reads any more.  Then a wave walks its positions with a compact loop over the taps: one
    //      accumulator, rows and tap blocks addressed dynamically.  (A first version kept the rows in registers and unrolled
    //      positions x taps: 96 KiB of code, beyond the 64 KiB instruction cache -- 36 us for this phase instead of 5.)
    const int tb0 = P0 - PL3 > 0 ? P0 - PL3 : 0, tb1 = P1 - 1 + PR3 < L1 - 1 ? P1 - 1 + PR3 : L1 - 1;    // rows [tb0, tb1]
    f4* stage_a = reinterpret_cast<f4*>(lut_s + 2048);      // LP_ROWS_A rows behind the LUT / flags / byte rows
    f4* stage_b = reinterpret_cast<f4*>(smem);              // further rows over the conv2 blocks
    {
        const int nrow = tb1 - tb0 + 1;
        const f4* src = o2g + (size_t)tb0 * 2 * 64;
        for (int i0 = tid; i0 < nrow * 128; i0 += 8 * 256) {
            f4 v[8];
            const int last = nrow * 128 - 1;
            auto at = [&](int k) { const int i = i0 + k * 256; return src + (i <= last ? i : last); };
            fx_load16x8_agent(at(0), at(1), at(2), at(3), at(4), at(5), at(6), at(7), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 256;
                if (i < nrow * 128) {
                    const int r = i >> 7;
                    (r < p.lp_rows_a ? stage_a + r * 128 : stage_b + (r - p.lp_rows_a) * 128)[i & 127] = v[k];
                }
            }
        }
    }
    __syncthreads();
    if (p.lp_debug == 7) return;
    f4 gmax = splat4(0.f);
    if (h1 > h0) {
        const f4 bias3 = *reinterpret_cast<const f4*>(&cb[32 * FT + 16 * mo + 4 * g]);
        // two positions at a time: two independent accumulator chains (a chain of 152 dependent MFMAs leaves the pipe idle
        // between them); per output element the order stays (tap, input tile, k-step)
        for (int o = h0; o < h1; o += 2) {
            f4 acc0 = bias3, acc1 = bias3;
            const bool two = o + 1 < h1;                     // (wave-uniform)
            // taps of output o: j in [j_lo, j_hi]; output o + 1 reads row (o + 1) + j - PL3: the same rows, one tap earlier
            const int j_lo = PL3 - o > 0 ? PL3 - o : 0, j_hi = L1 - 1 - o + PL3 < K3 - 1 ? L1 - 1 - o + PL3 : K3 - 1;
            // walk the ROWS t2 = o + j - PL3 that either output reads: t2 in [o + j_lo - PL3, (two ? o + 1 : o) + PR3] within [0, L1)
            const int t_first = o + j_lo - PL3;
            int t_last = (two ? o + 1 : o) + PR3;
            if (t_last > L1 - 1) t_last = L1 - 1;
            (void)j_hi;
#pragma unroll 2
            for (int t2 = t_first; t2 <= t_last; ++t2) {
                const int r = t2 - tb0;                      // row of out2[t2] in the staged block (wave-uniform)
                const f4* xr = r < p.lp_rows_a ? stage_a + r * 128 : stage_b + (r - p.lp_rows_a) * 128;
                const f4 x0 = xr[lane], x1 = xr[64 + lane];
                const int ja = t2 + PL3 - o, jb = ja - 1;    // tap of this row for output o / o + 1
                if (ja >= 0 && ja < K3) {
                    const f4 a0 = w_c3[((ja * FT + 0) * FT + mo) * 64 + lane];
                    const f4 a1 = w_c3[((ja * FT + 1) * FT + mo) * 64 + lane];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) acc0 = mfma16(a0[r4], x0[r4], acc0);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) acc0 = mfma16(a1[r4], x1[r4], acc0);
                }
                if (two && jb >= 0 && jb < K3) {
                    const f4 b0 = w_c3[((jb * FT + 0) * FT + mo) * 64 + lane];
                    const f4 b1 = w_c3[((jb * FT + 1) * FT + mo) * 64 + lane];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) acc1 = mfma16(b0[r4], x0[r4], acc1);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) acc1 = mfma16(b1[r4], x1[r4], acc1);
                }
            }
            gmax = pool_max4(gmax, acc0);
            if (two) gmax = pool_max4(gmax, acc1);
        }
        // The blocks of a unit meet in a zeroed pool through atomicMax on the float bits, as the SEG form's do -- but 116
        // blocks x 4 waves hitting the same 512 words serialise at the memory side (~85 ns per same-address atomic: 20 us of
        // a 38 us launch, profiles/r4_lp_stages.log).  So block b uses sub-pool b mod LP_POOLS (16x fewer contenders per
        // word; the head folds the sub-pools), and only lanes that hold a real sequence take part.
        if (live) {
            unsigned* pl = p.pool + ((((unit * LP_POOLS + (b % LP_POOLS)) * 2 + mo) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicMax(&pl[r], __float_as_uint(gmax[r]));
        }
    }
    if (p.lp_debug == 5) return;
    // (a bad character is reported BEFORE the block takes its ticket, and performed system-wide: the host may read the error word
    //  as soon as the last unit's completion flag is up)
    if (bad) { fx_raise(p.err, FX_ERR_BADCHAR); __threadfence_system(); }
    // ---- the blocks of a unit meet in the zeroed pool; the last to arrive runs the dense head (as the SEG form)
    __syncthreads();                                          // (every wave's atomicMax has been performed: vmcnt(0); device-scope atomics need no fence)
    if (p.lp_debug == 6) return;
    // Every block brings the head's weights (dense blocks + vectors, ~54 KiB) into LDS -- the conv blocks are not needed any more --
    // while its ticket makes the round trip to L2: the block that turns out to be the unit's last finds them in place instead of
    // starting a 2.5 us fill after it knows (the others are finished anyway; their fills cost L2 reads nobody is waiting for).
    unsigned ticket = 0;
    if (tid == 0) ticket = atomicAdd(&p.cnt[unit], 1u);
    const int head_floats = p.lp_head_floats;
    fill_lds(reinterpret_cast<f4*>(smem), reinterpret_cast<const f4*>(p.w[m] + p.off_d1), head_floats / 4);
    if (tid == 0) flags[1] = (ticket == (unsigned)NB - 1u) ? 1 : 0;
    __syncthreads();
    if (flags[1]) {
        // fold the unit's LP_POOLS sub-pools (2 KiB each): thread t takes 16-byte word t mod 128 of eight of them -- all eight
        // loads in flight at once, past the non-coherent cache levels -- and puts the entries back to zero; the two halves meet
        // in LDS (non-negative floats order like their bits: integer max)
        f4* fold = reinterpret_cast<f4*>(smem + ((head_floats + 3) & ~3));      // [2 halves][128 words], behind the head's weights
        f4* hx = fold + 256;                                                   // the head's exchange tiles: [HT] dense 1, [HT] dense 2
        {
            f4* base = reinterpret_cast<f4*>(p.pool) + (size_t)unit * LP_POOLS * 128 + (size_t)(tid >> 7) * 8 * 128 + (tid & 127);
            f4 v[8];
            fx_load16x8_agent(base, base + 128, base + 256, base + 384, base + 512, base + 640, base + 768, base + 896, v);
            f4 mx = v[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) mx = pool_max4(mx, v[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) fx_store16_agent(base + k * 128, splat4(0.f));
            fold[tid] = mx;
            fx_wait_vm(0);
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&p.cnt[unit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The dense head over all four waves (round 4; one wave ran its 252 MFMAs alone before: 2.6 us of a 7 us head): wave w owns
        // the output tiles {w, w + 4} of both layers, the tiles change hands through LDS -- the quad kernel's phases D / E / F.
        // Per output tile the same operands in the same order as pair_dense_head: the same bits.
        const f4* w_d1 = reinterpret_cast<const f4*>(smem);
        const f4* w_d2 = reinterpret_cast<const f4*>(smem + (p.off_d2 - p.off_d1));
        const float* db = smem + (p.off_db - p.off_d1);
        {
            f4 pooled[2];
            pooled[0] = pool_max4(fold[lane], fold[128 + lane]);
            pooled[1] = pool_max4(fold[64 + lane], fold[192 + lane]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int to = wave + 4 * k;
                if (to < HT) {
                    f4 acc = *reinterpret_cast<const f4*>(&db[16 * to + 4 * g]);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const f4 a = w_d1[(mi * HT + to) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = mfma16(a[r], pooled[mi][r], acc);
                    }
                    hx[to * 64 + lane] = relu4(acc);
                }
            }
        }
        __syncthreads();
        {
            f4 h1v[HT];
#pragma unroll
            for (int mi = 0; mi < HT; ++mi) h1v[mi] = hx[mi * 64 + lane];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int to = wave + 4 * k;
                if (to < HT) {
                    f4 acc = *reinterpret_cast<const f4*>(&db[16 * HT + 16 * to + 4 * g]);
#pragma unroll
                    for (int mi = 0; mi < HT; ++mi) {
                        const f4 a = w_d2[(mi * HT + to) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (mi == HT - 1 && r >= p.rlh) break;
                            acc = mfma16(a[r], h1v[mi][r], acc);
                        }
                    }
                    hx[(HT + to) * 64 + lane] = relu4(acc);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            f4 h2v[HT][1];
#pragma unroll
            for (int mi = 0; mi < HT; ++mi) h2v[mi][0] = hx[(HT + mi) * 64 + lane];
            float y[1];
            final_dot<HT, 1>(db + 32 * HT, db[48 * HT], h2v, y, g);
            if (g == 0 && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y[0]);
            if (p.lp_done) {
                // completion flag: this unit's scores performed system-wide (they may live in pinned host memory), then the count of
                // finished units; the last one puts the counter back and raises the flag the host is polling
                __threadfence_system();
                if (lane == 0) {
                    unsigned* done = p.lp_bar + 32 * 17;
                    const unsigned t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (t + 1u == (unsigned)(p.M * p.TG)) {
                        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p.lp_done, p.lp_done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    }
}

// Launches the LP form when it applies: FX_EUNSUPPORTED otherwise (the caller carries on with the SEG / whole-sequence forms).
template <int A, int K, int HT>
int launch_lp(fx_engine* e, PairArgs a, size_t lds_bytes) {
    constexpr int PBW = 8, K3 = A - 1;
    const int64_t U = (int64_t)a.M * a.TG;
    const int L1 = a.L - K + 1;
    // LDS: the conv image, 2 KiB of LUT / flags / byte rows, then staged out2 rows up to the CU's limit; more rows over the
    // conv2 blocks.  A block of PB positions stages PB + A - 2 rows.
    const size_t image = (size_t)a.lds_floats * 4 + 2048;
    if (image + 4096 > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    lds_bytes = (size_t)e->max_lds;
    const int rows_a = (int)((lds_bytes - image) / 2048);
    const int rows_b = (int)(((size_t)(a.off_c3 - a.lds_from) * 4) / 2048);
    int pb_max = rows_a + rows_b - (K3 - 1);               // positions per block the staging area allows
    if (pb_max > 2 * PBW) pb_max = 2 * PBW;
    if (pb_max < 2) return FX_EUNSUPPORTED;
    int64_t room = e->num_cus - 8;                         // every workgroup must find a CU at once (the barrier); a few stay free
    const int64_t nb_min = (L1 + pb_max - 1) / pb_max;      // <= PBW positions per wave (two halves per block), rows that fit the staging area
    if (!e->cnn_lp || L1 < 24 || U < 1 || U * nb_min > room || a.lp_head_floats <= 0 || (size_t)a.lp_head_floats * 4 > lds_bytes) return FX_EUNSUPPORTED;
    // resident scoring workgroups of another ensemble hold most of their CU's LDS: work beside them when there is room for
    // a useful grid, else tell them to leave (they do within microseconds; the barrier simply waits for the CUs they free)
    if (e->server.running) {
        if (U * nb_min * 2 <= room - e->server.wgs) room -= e->server.wgs;
        else fx_server_stop(e);
    }
    int64_t nb = room / U;
    if (nb > L1 / 2) nb = L1 / 2;                          // >= 2 positions per block: one per half
    if (nb < nb_min) return FX_EUNSUPPORTED;
    auto kern = k_score_cnn_lp<A, K, HT>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    void* ws = nullptr;
    const size_t pool_bytes = (size_t)U * LP_POOLS * 2 * 64 * 4 * sizeof(unsigned), cnt_bytes = (size_t)U * sizeof(unsigned);
    if (int rc = fx_zero_pool(e, pool_bytes + cnt_bytes, &ws)) return rc;
    a.pool = (unsigned*)ws;
    a.cnt = (unsigned*)((char*)ws + pool_bytes);
    void* o2 = nullptr;
    if (int rc = fx_scratch(e, 2, (size_t)U * L1 * 2 * 64 * sizeof(f4), &o2)) return rc;
    a.lp_out2 = (f4*)o2;
    if (!e->d_lp_bar) {
        if (hipMalloc(reinterpret_cast<void**>(&e->d_lp_bar), FX_LP_BAR_BYTES) != hipSuccess) { (void)hipGetLastError(); return fx_fail(e, FX_ENOMEM, "hipMalloc of the barrier counters failed"); }
        FX_HIP(e, hipMemsetAsync(e->d_lp_bar, 0, FX_LP_BAR_BYTES, e->stream));
        for (unsigned& t : e->lp_bar_total) t = 0;
    }
    a.lp_bar = e->d_lp_bar;
    a.lp_nb = (int)nb;
    a.lp_rows_a = rows_a;
    a.lp_debug = (int)e->cnn_lp_debug;
    a.lp_done = nullptr;
    a.lp_mail = nullptr;
    if (e->done_flag && !a.lp_debug) {
        if (++e->done_seq == 0) ++e->done_seq;
        if (e->done_seq >= 0x7FFFFFFFu) e->done_seq = 1;   // (a pre-launched instance reports (sequence << 1) | 1)
        a.lp_done = e->d_done; a.lp_done_seq = e->done_seq;
        e->done_armed = true;
        if (e->lp_arm_next && e->lp_mail && e->d_lp_state) {
            a.lp_mail = e->lp_mail;
            a.ascii = e->lp_mail->bytes;
            a.lp_word = ((unsigned long long)e->done_seq << 16) | (unsigned long long)a.N;
            a.lp_idle_ticks = (unsigned long long)e->serve_idle_us * 100ull;
            a.lp_state = e->d_lp_state;
            a.lp_decide = e->d_lp_bar + 18 * 32;
        }
    }
    e->lp_launches += 1;
    const int64_t G = U * nb;
    const bool arrives = !(a.lp_debug >= 1 && a.lp_debug <= 3);   // (profiling stages that leave before the barrier do not arrive at it)
    for (int gi = 0; gi < 16; ++gi) {
        const unsigned members = (unsigned)((G - gi + 15) / 16);  // blocks b with b mod 16 == gi
        if (arrives) {
            e->lp_bar_total[1 + gi] += members;
            if (members) e->lp_bar_total[0] += 1;
        }
        a.lp_gtarget[gi] = e->lp_bar_total[1 + gi];
    }
    a.lp_target = e->lp_bar_total[0];
    e->lp_launched = true;
    hipLaunchKernelGGL(kern, dim3((unsigned)(U * nb)), dim3(256), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

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
    // Small batch (a CMA-ES / DyNA-PPO population, a single sequence): fewer units than half the CUs.  Cut every
    // tile into segments over SB workgroups so that the call's latency is ~L1/S + halo steps.  The halo (PL3 + PL2 +
    // PR2 + PR3 = 22 positions at A = 20) is recomputed by every segment, but the machine is otherwise empty: as many
    // workgroups per tile as fit in ONE wave of the grid (U x SB <= CUs), down to segments of two positions
    // (tools/archive/runs/r2_pair_seg_sweep.py: a 1-16 sequence call at L = 237 250 -> 181 us, at L = 90 269 -> 169 us).
    int64_t sb = fx_pair_seg_count(L1, WAVES / 2, U, e->num_cus);
    if (e->cnn_pair_seg == 0 || 2 * U > e->num_cus) sb = 0;
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
    if (e->cnn_pair_seg == 0 || 2 * U > e->num_cus) sb = 0;
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
