// K2a (round 6): the first layer of an MLP whose kernel rows do not fit LDS, POSITION-MAJOR.  Part of score_dense_mfma.hip (included there).
//
// mlp.py:21-31 on a protein landscape -- DyNA-PPO's default member MLP(seq_len, 200, alphabet), dyna_ppo.py:54, on AAV's 90 residues: the
// first layer is seq_len x 20 rows of H floats, 1.5 MB.  As a per-sequence gather (k_score_dense_mfma, W1G) every sequence pulls its seq_len
// rows -- 75 KB -- from L2: 7.5 GB for 1e5 sequences, 820 us, 0.13 of the f32-MFMA rate the two H x H layers could run at; 3.5 ms at 237
// residues (profiles/r6_protein_mlp_before.log).  Here the loop is turned around: a workgroup holds the first-layer accumulators of up to
// 8 x TW tiles in registers and walks the POSITIONS; the A rows of a position (17 KB) are copied global -> LDS once per workgroup (direct
// copies, double-buffered, KP positions per barrier) and every wave adds the row its sequence's letter selects: the rows cross L2 -> LDS once
// per 8 x TW tiles instead of once per sequence, and the gather runs at LDS bandwidth.  The relu'd sums go to a scratch buffer in the
// B-operand layout the next layer's MFMAs want; k_score_dense_mfma then runs with `h1` set and skips its own first layer.
// Every element is bias + rows in position order, as in the gather form: the SAME BITS.
#pragma once
#include "fx_common.h"
#include "mfma_common.h"

namespace {

struct L1Args {
    const uint8_t* ascii;       // N x L
    const uint8_t* lut;
    const float* w[FX_MAX_M];   // packed weights per member
    f4* h1;                     // [(member * TG + tile) * HT + mo][64 lanes]
    unsigned* err;
    int64_t N, TG;
    int M, L, A;
    int off_w1p, off_db;
};

// LDS: LUT (256 B), then two slabs of KP x A rows.
// Lanes: the gather is 16 sequences reading 16 DIFFERENT rows at once, which no layout keeps off each other's banks (sixteen random
// letters on sixteen 16-byte slots: ~3 LDS cycles per access instead of 1).  So the four lanes of a sequence are NEIGHBOURS here --
// lane = 4 sq + g, not the MFMA layout's 16 g + sq: a ds_read_b128 is served four lanes x four sequences at a time, a sequence's lanes
// read 64 contiguous bytes of its row, and with rows RS floats apart, RS mod 64 = 16 or 48, the row of letter c starts in 64-byte window
// (c mod 4) of the 256-byte bank row -- four random letters on four windows: ~2 cycles.  The sums are stored to the scratch at the MFMA
// layout's lane index, so the permutation costs nothing.
template <int HT> struct FxL1Row { static constexpr int R = (16 * HT) % 64, PAD = (R == 16 || R == 48) ? 0 : (16 - R + 64) % 64, RS = 16 * HT + PAD; };
template <int HT, int TW, int KP, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_mlp_l1_pos(L1Args p) {
    constexpr int RS = FxL1Row<HT>::RS;
    static_assert(KP == 2 || KP == 4, "a slab's bytes are one 2- or 4-byte load per sequence");
    static_assert(RS % 64 == 16 || RS % 64 == 48, "rows of consecutive letters start in consecutive 64-byte windows");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane & 3, sq = lane >> 2, lane_out = 16 * g + sq;      // (see above)
    const int L = p.L, A = p.A;
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(smem);
    float* rows = smem + 64;
    const int slab_floats = KP * A * RS;
    const unsigned rows_lds = __builtin_amdgcn_readfirstlane(fx_lds_addr(rows));
    for (int i = tid; i < 64; i += WAVES * 64)
        reinterpret_cast<uint32_t*>(lut_s)[i] = reinterpret_cast<const uint32_t*>(p.lut)[i];
    int64_t u_lo, u_hi;
    fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    const unsigned amax = (unsigned)A - 1u;
    const int NS = (L + KP - 1) / KP;
    bool bad = false;
    for (int m = m_first; m <= m_last; ++m) {
        const float* w1p = p.w[m] + p.off_w1p;
        const float* db = p.w[m] + p.off_db;
        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        // the rows of positions [s KP, s KP + KP) -> buffer `buf`: one 16-byte copy per lane and row (4 HT lanes), rows dealt to the waves
        auto issue = [&](int s, int buf) {
            const int r_lo = s * KP * A;
            const int r_hi = ((s + 1) * KP < L ? (s + 1) * KP : L) * A;
            for (int r = r_lo + wave; r < r_hi; r += WAVES) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(rows_lds + (unsigned)((buf * slab_floats + (r - r_lo) * RS) * 4));
                if (lane < 4 * HT) fx_dma16(w1p + (size_t)r * (16 * HT) + lane * 4, dst);
            }
        };
        for (int64_t base = t_lo; base < t_hi; base += WAVES * TW) {
            // this wave's tiles of the pass: base + wave, + WAVES, ... (a short last pass spreads over the waves)
            bool live[TW];
            const uint8_t* row[TW];
            int64_t tile[TW];
#pragma unroll
            for (int k = 0; k < TW; ++k) {
                tile[k] = base + wave + (int64_t)WAVES * k;
                live[k] = tile[k] < t_hi;
                const int64_t n = tile[k] * 16 + sq;
                row[k] = p.ascii + ((live[k] && n < p.N) ? n : 0) * L;     // (lanes past the batch recompute sequence 0)
            }
            f4 acc[TW][HT];
#pragma unroll
            for (int k = 0; k < TW; ++k)
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) acc[k][mo] = *reinterpret_cast<const f4*>(&db[16 * mo + 4 * g]);
            // the KP bytes of slab s of this lane's sequences, one load per tile (the last, partial slab: byte by byte)
            auto load_bytes = [&](int s, unsigned (&out)[TW]) {
#pragma unroll
                for (int k = 0; k < TW; ++k) {
                    unsigned v = 0;
                    if (live[k]) {
                        const uint8_t* q = row[k] + s * KP;
                        if (s * KP + KP <= L) {
                            if constexpr (KP == 2) { unsigned short t; __builtin_memcpy(&t, q, 2); v = t; }
                            else __builtin_memcpy(&v, q, 4);
                        } else {
                            for (int b = 0; s * KP + b < L; ++b) v |= (unsigned)q[b] << (8 * b);
                        }
                    }
                    out[k] = v;
                }
            };
            unsigned seen = 0, cur[TW], nxt[TW];
            __syncthreads();                                   // the previous pass's readers are done with both buffers
            issue(0, 0);
            load_bytes(0, cur);
            for (int s = 0; s < NS; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies of slab s (and its bytes) have landed ...
                __syncthreads();                                   // ... everybody's have; the other buffer is free
                if (s + 1 < NS) {
                    issue(s + 1, (s + 1) & 1);
                    load_bytes(s + 1, nxt);
                }
                const float* buf = rows + (s & 1) * slab_floats + 4 * g;
                // the slab's letters first (KP x TW byte look-ups in flight together), then the gathers: a row's address no longer waits
                // for its own look-up in front of every gather (two waves per SIMD hide little)
                int off[KP][TW];
#pragma unroll
                for (int pp = 0; pp < KP; ++pp)
#pragma unroll
                    for (int k = 0; k < TW; ++k) {
                        const unsigned c = lut_s[(cur[k] >> (8 * pp)) & 0xFFu];
                        if (live[k] && s * KP + pp < L) seen |= c;      // a code is < A <= 127, or 0xFF: tested once per pass
                        const unsigned ci = c < amax ? c : amax;
                        off[pp][k] = (pp * A + (int)ci) * RS;
                    }
#pragma unroll
                for (int pp = 0; pp < KP; ++pp) {
                    if (s * KP + pp < L) {
#pragma unroll
                        for (int k = 0; k < TW; ++k) {
                            if (live[k]) {
                                const float* rowp = buf + off[pp][k];
#pragma unroll
                                for (int mo = 0; mo < HT; ++mo) acc[k][mo] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < TW; ++k) cur[k] = nxt[k];
            }
            bad |= seen >= 0x80u;
#pragma unroll
            for (int k = 0; k < TW; ++k) {
                if (live[k]) {
                    f4* dst = p.h1 + (((int64_t)m * p.TG + tile[k]) * HT) * 64 + lane_out;
#pragma unroll
                    for (int mo = 0; mo < HT; ++mo) dst[mo * 64] = relu4(acc[k][mo]);
                }
            }
        }
    }
    if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

template <int HT, int TW, int KP, int WAVES = 8>
int launch_l1_pos(fx_engine* e, const L1Args& a) {
    auto kern = k_mlp_l1_pos<HT, TW, KP, WAVES>;
    const size_t lds = 256 + (size_t)2 * KP * a.A * FxL1Row<HT>::RS * 4;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    const int64_t U = (int64_t)a.M * a.TG;
    if (blocks > U) blocks = U;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

// FX_EUNSUPPORTED: no instantiation for this hidden width / the slabs do not fit LDS (the caller keeps the gather form)
inline int fx_launch_mlp_l1_pos(fx_engine* e, const L1Args& a, int HT) {
    switch (HT) {
        case 1: return launch_l1_pos<1, 4, 4>(e, a);
        case 2: return launch_l1_pos<2, 4, 4>(e, a);
        case 4: return launch_l1_pos<4, 4, 4>(e, a);
        case 7: return launch_l1_pos<7, 2, 4, 16>(e, a);         // (four waves per SIMD, two tiles each: -7 % against 8 waves x 4 tiles)
        case 8: return launch_l1_pos<8, 4, 4>(e, a);
        case 13: {
            // three waves per SIMD, two tiles each, four positions per barrier where two slabs of them fit (A <= 23); else two per barrier
            const int rc = launch_l1_pos<13, 2, 4, 12>(e, a);
            return rc != FX_EUNSUPPORTED ? rc : launch_l1_pos<13, 3, 2>(e, a);
        }   // (four positions per barrier where two slabs of them fit: A <= 23)
        case 16: return launch_l1_pos<16, 2, 2>(e, a);
        default: return FX_EUNSUPPORTED;
    }
}

}  // namespace

// ---- host-side decisions about the MLP's first layer (shared with score_dense_small.hip and the planner in fx_score.hip)
// Which terms the persistent kernel's MLP first layer adds (the small-launch kernel must add the same ones):
// 0 = one kernel row per position, 1 = one pre-summed row per PAIR of positions (the PAIR form), 2 = MFMA form (A/B option).
int fx_mlp_first_layer_form(fx_engine* e, const FxShape& s, const FxPackLayout& lay) {
    if (e->mlp_l1_mfma) return 2;
    if (!e->mlp_pair || lay.off_w1pair < 0) return 0;
    if (lay.HT > 8) {
        // H > 128 (round 6): the slab form with the pair rows in LDS -- nothing else of the image is (the vectors are read from L2) -- beside
        // its two slabs; mlp_pair = 2 keeps the plain rows there (A/B)
        if (!e->dense_slab || e->mlp_pair == 2) return 0;
        const size_t need_slab = (size_t)lay.pair_floats * 4 + 256 + 32 + (size_t)2 * FX_SLAB_KG * lay.HT * 1024;
        return need_slab <= (size_t)e->max_lds ? 1 : 0;
    }
    const size_t need = (size_t)(lay.total_floats - lay.off_d2 + lay.pair_floats) * 4 + 256 + 32;
    return need <= (size_t)e->max_lds ? 1 : 0;
}

bool fx_mlp_l1_pos_applies(const fx_engine* e, const FxShape& s, const FxPackLayout& lay) {
    if (s.kind != FX_MLP || !e->mlp_l1_pos || e->mlp_l1_mfma || s.A > 127) return false;
    return (size_t)((lay.HT > 8 ? lay.off_d2 : lay.total_floats) - lay.off_w1p) * 4 + 256 + 32 > (size_t)e->max_lds;
}

