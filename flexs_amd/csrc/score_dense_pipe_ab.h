// The software-pipelined MLP / GlobalEpistasis form of round 3 (engine option dense_pipe): measured 11-13 % slower than the 16-wave form of
// score_dense_mfma.hip at every size (profiles/r3_dense_pipe_ab.log), so it is compiled into the A/B build only (`make ab`, -DFX_AB).
// Included by score_dense_mfma.hip inside its anonymous namespace, after DenseArgs and k_score_dense_mfma (it shares their helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form (round 3) of the two LDS-resident first-layer forms above: PAIR (MLP, 4-letter alphabet)
// and BT (GlobalEpistasis byte table).
//
// In the form above a wave runs a tile's layers one after the other: the first layer is an LDS gather with ~3 dependent
// LDS round trips per position and no MFMA at all, the hidden layers are MFMA chains.  At 1e5 sequences (6 tiles per
// SIMD) all waves of a CU start in the gather together (7 us with sixteen waves on one LDS, matrix pipes idle) and keep
// meeting there; PMC: pipe busy 0.49, 21 % of LDS cycles in bank conflicts.  Here the first layer of tile
// t + 1 runs INSIDE the hidden layers of tile t: its 16 x L bytes are requested from global memory before tile t's first
// MFMA, parked in the wave's LDS scratch during the first block row, and one pair-row gather (MLP) / one 32-position
// table trip (GE) is issued in front of each later block row's MFMA cluster and added behind it -- the LDS latency is
// covered by ~28 MFMAs of the same wave, and no wave ever sits in a gather-only phase after its first tile.  The
// next tile's first-layer sums (7 f4 for H = 100) wait in registers, which is why this form runs 8 waves per
// workgroup (256-VGPR budget, two waves per SIMD) instead of 16.  Start-up is cut the same way: the weight image comes
// by direct global -> LDS copies in two parts (mfma_common.h fx_dma_fill) -- what the first layer reads (pair rows /
// byte table + vectors) first, the H x H blocks behind it -- and the first tile's gather starts when part one has
// landed.
// Every output element sees the arithmetic of the form above -- bias + rows in position order, (input tile, k-step)
// order with the same tail skip, the same dot -- so the scores are the SAME BITS (GPU test).
// One action per block row: the sequence must fit (L / 2 + 5 <= 2 HT for the MLP, Lpad / 32 + 3 <= HT for GE); longer
// sequences and narrower hidden layers keep the form above.
template <int KIND, int HT, int NLD, int VAR>
__global__ void __launch_bounds__(512) k_score_dense_pipe(DenseArgs p) {
    // VAR 1: the A operands of block row i + 1 are requested by hand in front of row i's MFMAs and a scheduling barrier keeps the
    // front loads in front; VAR 0: the rows' loads are written where they are consumed and the compiler places them
    constexpr int WAVES = 8;
    constexpr bool MLP = KIND == FX_MLP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, sq = lane & 15;
    const int L = p.L;
    float* img = smem + (MLP ? 0 : p.Lpad * 32);
    float* wpair = img + p.lds_floats;
    float* aux = wpair + (MLP ? p.pair_floats : 0);
    uint8_t* lut_s = reinterpret_cast<uint8_t*>(aux);
    int* next_tile = reinterpret_cast<int*>(aux + 64);
    int* simd_waves = next_tile + 4;
    uint8_t* stw = reinterpret_cast<uint8_t*>(aux + 64 + 8) + (tid >> 6) * p.stage_stride;
    fx_stamp(p.trace, 0);
    if (p.wave_prio) fx_stagger_priority();
    const int simd = fx_simd_id();
    fx_stamp(p.trace, 7, (unsigned long long)simd + 1);
    if (tid < 4) simd_waves[tid] = 0;

    int64_t u_lo, u_hi;
    fx_unit_range(p.TG, p.M, u_lo, u_hi);
    if (u_lo >= u_hi) return;
    const int m_first = (int)(u_lo / p.TG), m_last = (int)((u_hi - 1) / p.TG);
    bool bad = false;
    [[maybe_unused]] unsigned tiles_done = 0;
    FxSimdShare share{0, 1, 1};
    if (!MLP && lane < 32) stw[16 * L + lane] = (uint8_t)p.bt_base;       // what the padded trips of a tile's last row read
    const int np2 = L >> 1;
    // actions of a pipelined first layer.  MLP: park, index, one per pair (+ odd tail).  GE: park, then the trips, each
    // spread over two slots (bytes, then table reads + adds): the last trip's adds land in the last slot
    const int nact = MLP ? 2 + np2 + (L & 1) : 2 + p.Lpad / 32;
    const int vec_from = MLP ? p.off_db - p.lds_from : 0;                 // part one of the image: the vector area (MLP: at its end)

    for (int m = m_first; m <= m_last; ++m) {
        __syncthreads();
        if (tid < 4) next_tile[tid] = 0;
        if (m == m_first) fx_count_simd_wave(simd_waves, simd);
        // ---- weights: part one (first layer) and part two (H x H blocks) by direct global -> LDS copies
        int part2 = 0;
        {
            const float* src = p.w[m] + p.lds_from;
            if (m == m_first) fx_lut_dma(lut_s, p.lut);
            if (MLP) {
                fx_dma_fill(img + vec_from, src + vec_from, (p.lds_floats - vec_from) / 4, WAVES);
                fx_dma_fill(wpair, p.w[m] + p.off_w1pair, p.pair_floats / 4, WAVES);
                part2 = fx_dma_fill(img, src, vec_from / 4, WAVES);
            } else {
                // GE image = [d3 blocks][vectors]: vectors + byte table first
                const int blk = p.off_db - p.lds_from;
                fx_dma_fill(img + blk, src + blk, (p.lds_floats - blk) / 4, WAVES);
                fx_dma_fill(smem, p.bt[m], p.Lpad * 8, WAVES);
                part2 = fx_dma_fill(img, src, blk / 4, WAVES);
            }
        }
        fx_wait_vm(part2);                                   // this wave's share of part one has landed
        __syncthreads();
        if (m == m_first) share = fx_simd_share(simd_waves, simd);
        if (m == m_first) fx_stamp(p.trace, 1);
        const f4* w_d2 = reinterpret_cast<const f4*>(img + (p.off_d2 - p.lds_from));
        const f4* w_d3 = reinterpret_cast<const f4*>(img + (p.off_d3 - p.lds_from));
        const float* db = img + (p.off_db - p.lds_from);

        const int64_t t_lo = (u_lo > (int64_t)m * p.TG ? u_lo : (int64_t)m * p.TG) - (int64_t)m * p.TG;
        const int64_t t_hi = (u_hi < (int64_t)(m + 1) * p.TG ? u_hi : (int64_t)(m + 1) * p.TG) - (int64_t)m * p.TG;
        const int64_t s_lo = t_lo + (t_hi - t_lo) * share.before / share.total;
        const int64_t s_hi = t_lo + (t_hi - t_lo) * (share.before + share.mine) / share.total;
        auto pull = [&]() -> int64_t {
            int got = 0;
            if (lane == 0) got = atomicAdd(&next_tile[simd], 1);
            got = __builtin_amdgcn_readfirstlane(got);
            return s_lo + got < s_hi ? s_lo + got : -1;
        };
        auto rows_of = [&](int64_t tg) -> int { return (int)(p.N - tg * 16 < 16 ? p.N - tg * 16 : 16); };

        // ---- first layer, whole (the wave's first tile, and a ragged last tile of the batch): as in the form above
        f4 hN[MLP ? HT : 1];                                // MLP: bias + gathered rows of the NEXT tile
        float sN = 0.f;                                     // GE: this lane group's partial sum of the NEXT tile
        unsigned seenN = 0;
        auto l1_full = [&](int64_t tg) {
            const int rows = rows_of(tg);
            fx_stage_tile(p.ascii + tg * 16 * L, rows * L, stw, lane);
            const uint8_t* srow = stw + (tg * 16 + sq < p.N ? sq : 0) * L;
            if constexpr (MLP) {
                f4 (&h)[HT] = reinterpret_cast<f4 (&)[HT]>(hN);
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = *reinterpret_cast<const f4*>(&db[16 * mo + 4 * g]);
                fx_lds_u8p rb = (fx_lds_u8p)srow;
                for (int p0 = 0; p0 < np2; ++p0) {
                    const unsigned c0 = lut_s[rb[2 * p0]], c1 = lut_s[rb[2 * p0 + 1]];
                    seenN |= c0 | c1;
                    const float* rowp = wpair + (p0 * 16 + (((c0 & 3u) << 2) | (c1 & 3u))) * (16 * HT + FX_PAIR_PAD) + 4 * g;
#pragma unroll
                    for (int mo = 0; mo < HT; ++mo) h[mo] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                }
                if (L & 1) {
                    const unsigned c0 = lut_s[rb[L - 1]];
                    seenN |= c0;
                    const float* rowp = wpair + (np2 * 16 + (c0 & 3u)) * (16 * HT + FX_PAIR_PAD) + 4 * g;
#pragma unroll
                    for (int mo = 0; mo < HT; ++mo) h[mo] += *reinterpret_cast<const f4*>(rowp + 16 * mo);
                }
            } else {
                const char* tb = reinterpret_cast<const char*>(smem) + g * 128 - p.bt_base * 4;
                const bool check = p.validate && (int)(tg % p.M) == m;
                fx_lds_u8p rp = (fx_lds_u8p)(srow + g);
                unsigned seen = 0;
                sN = 0.f;
                if (rows == 16) {
                    for (int t = 0; t < p.Lpad; t += 32) {
                        int raw[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) raw[k] = rp[t + 4 * k];
#pragma unroll
                        for (int k = 0; k < 8; ++k) sN += *reinterpret_cast<const float*>(tb + t * 128 + k * 512 + raw[k] * 4);
                        if (check) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) seen |= (t + 4 * k + g < L) ? (unsigned)lut_s[raw[k]] : 0u;
                        }
                    }
                } else {
                    for (int l = g; l < L; l += 4) {
                        const int raw = rp[l - g];
                        sN += *reinterpret_cast<const float*>(tb + (l - g) * 128 + raw * 4);
                        seen |= lut_s[raw];
                    }
                    if (!check) seen = 0;
                }
                seenN |= seen;
            }
        };
        // ---- first layer, pipelined: action `a` of `nact` for the (full) tile whose bytes wait in `rb`
        FxBytes16 rb[NLD];
        auto l1_request = [&](int64_t tg) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int off = (lane + 64 * k) * 16;
                if (off < 16 * L) rb[k] = *reinterpret_cast<const FxBytes16*>(p.ascii + tg * 16 * L + off);
            }
        };
        auto l1_park = [&]() {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int off = (lane + 64 * k) * 16;
                if (off < 16 * L) {
                    uint32_t* d = reinterpret_cast<uint32_t*>(stw + off);
                    d[0] = rb[k].w[0]; d[1] = rb[k].w[1]; d[2] = rb[k].w[2]; d[3] = rb[k].w[3];
                }
            }
            if constexpr (MLP) {
                f4 (&h)[HT] = reinterpret_cast<f4 (&)[HT]>(hN);
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = *reinterpret_cast<const f4*>(&db[16 * mo + 4 * g]);
            } else {
                sN = 0.f;
            }
        };

        int64_t t_cur = pull();
        if (t_cur >= 0) {
            fx_stamp(p.trace, 2);
            l1_full(t_cur);
        }
        fx_wait_vm(0);                                       // part two (this wave's share) ...
        __syncthreads();                                     // ... and everybody else's
        // Schedule of the next tile's first layer over the block rows ("slots") of this tile's H x H layer(s); the last
        // action lands in the last slot, the bytes are requested at the top of the tile and parked `start` slots later
        // (an L2 round trip is ~1 us, a block row ~0.4 us).
        //   MLP  slots 0 .. 2 HT - 1:  park | all pair indices (bytes + LUT -> 4 bits per pair, packed) | one pair row per slot [| odd tail]
        //   GE   slots 0 .. HT - 1:    park | trip k: bytes in slot s, table reads in front of slot s + 1's MFMAs, adds behind them
        const int nslots = MLP ? 2 * HT : HT;
        const int start = nslots - nact;                     // (the launcher guarantees start >= 1)
        while (t_cur >= 0) {
            const int64_t tg = t_cur;
            const int64_t t_nxt = pull();
            const bool pipe = t_nxt >= 0 && rows_of(t_nxt) == 16;
            if (pipe) l1_request(t_nxt);
            asm volatile("" ::: "memory");
            const int64_t n = tg * 16 + sq;
            bad |= seenN >= 0x80u;
            seenN = 0;
            const uint8_t* srowN = stw + sq * L;             // (a pipelined tile is full: every lane has its own row)
            f4 h[HT], h2[HT];
            float y;
            if constexpr (MLP) {
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = relu4(hN[mo]);
                unsigned long long pidx = 0;                 // 4 bits per pair of positions: the pair row's index
                unsigned tailc = 0;
                // The A operands (7 x 16 bytes per lane and block row) are double-buffered by hand: row i + 1's are requested in
                // front of row i's MFMAs (two waves per SIMD cannot hide an LDS round trip per block row the way four do).
                auto block_row = [&](f4 (&a)[HT], f4 (&an)[HT], const f4* wcur, const f4* wnext, int mnext, const f4 (&in)[HT], f4 (&acc)[HT], int mi, int slot) {
                    if constexpr (VAR == 1) {
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) an[mo] = wnext[(mnext * HT + mo) * 64 + lane];
                    } else {
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) a[mo] = wcur[(mi * HT + mo) * 64 + lane];
                    }
                    const int act = slot - start;            // wave-uniform
                    const bool gather = pipe && act >= 2;
                    f4 v[HT];
                    if (gather) {                            // requested in front of the row's MFMAs ...
                        const int pp = act - 2;
                        const unsigned idx = pp < np2 ? (unsigned)(pidx >> (4 * pp)) & 15u : tailc;
                        const float* rowp = wpair + (pp * 16 + idx) * (16 * HT + FX_PAIR_PAD) + 4 * g;
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) v[mo] = *reinterpret_cast<const f4*>(rowp + 16 * mo);
                    }
                    if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mi == HT - 1 && r >= p.rlh) break;
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) acc[mo] = mfma16(a[mo][r], in[mi][r], acc[mo]);
                    }
                    if (gather) {                            // ... added behind them
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) hN[mo] += v[mo];
                    }
                    if (pipe && act == 0) l1_park();
                    if (pipe && act == 1) {
                        fx_lds_u8p rbn = (fx_lds_u8p)srowN;
                        for (int pp = 0; pp < np2; ++pp) {
                            const unsigned c0 = lut_s[rbn[2 * pp]], c1 = lut_s[rbn[2 * pp + 1]];
                            seenN |= c0 | c1;
                            pidx |= (unsigned long long)(((c0 & 3u) << 2) | (c1 & 3u)) << (4 * pp);
                        }
                        if (L & 1) {
                            tailc = lut_s[rbn[L - 1]];
                            seenN |= tailc;
                            tailc &= 3u;
                        }
                    }
                };
                f4 a0[HT], a1[HT];
                if constexpr (VAR == 1) {
#pragma unroll
                    for (int mo = 0; mo < HT; ++mo) a0[mo] = w_d2[mo * 64 + lane];
                }
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h2[mo] = *reinterpret_cast<const f4*>(&db[16 * HT + 16 * mo + 4 * g]);
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) {            // (HT may be odd: the buffers alternate by the row's global index)
                    const f4* wn = mi + 1 < HT ? w_d2 : w_d3;
                    const int mn = mi + 1 < HT ? mi + 1 : 0;
                    if ((mi & 1) == 0 || VAR == 0) block_row(a0, a1, w_d2, wn, mn, h, h2, mi, mi);
                    else block_row(a1, a0, w_d2, wn, mn, h, h2, mi, mi);
                }
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h2[mo] = relu4(h2[mo]);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = *reinterpret_cast<const f4*>(&db[32 * HT + 16 * mo + 4 * g]);
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) {
                    const int mn = mi + 1 < HT ? mi + 1 : 0;  // (the last prefetch is harmless: row 0 again)
                    if (((HT + mi) & 1) == 0 || VAR == 0) block_row(a0, a1, w_d3, w_d3, mn, h2, h, mi, HT + mi);
                    else block_row(a1, a0, w_d3, w_d3, mn, h2, h, mi, HT + mi);
                }
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = relu4(h[mo]);
                float yy[1];
                f4 hh[HT][1];
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) hh[mo][0] = h[mo];
                final_dot<HT, 1>(db + 48 * HT, db[64 * HT], hh, yy, g);
                y = yy[0];
            } else {
                const bool checkN = pipe && p.validate && (int)(t_nxt % p.M) == m;
                float s = sN;
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                s += db[0];
                const float sv = relu1(s);
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) {
                    const f4 w2 = *reinterpret_cast<const f4*>(&db[4 + 16 * mo + 4 * g]);
                    const f4 b2 = *reinterpret_cast<const f4*>(&db[4 + 16 * HT + 16 * mo + 4 * g]);
                    f4 v;
                    v.x = relu1(fmaf(sv, w2.x, b2.x));
                    v.y = relu1(fmaf(sv, w2.y, b2.y));
                    v.z = relu1(fmaf(sv, w2.z, b2.z));
                    v.w = relu1(fmaf(sv, w2.w, b2.w));
                    h2[mo] = v;
                }
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = *reinterpret_cast<const f4*>(&db[4 + 32 * HT + 16 * mo + 4 * g]);
                const char* tb = reinterpret_cast<const char*>(smem) + g * 128 - p.bt_base * 4;
                fx_lds_u8p rp = (fx_lds_u8p)(srowN + g);
                int raw[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) raw[k] = 0;
                f4 ab[2][HT];
                if constexpr (VAR == 1) {
#pragma unroll
                    for (int mo = 0; mo < HT; ++mo) ab[0][mo] = w_d3[mo * 64 + lane];
                }
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) {
                    f4 (&a)[HT] = ab[VAR == 1 ? (mi & 1) : 0];
                    if constexpr (VAR == 1) {
                        if (mi + 1 < HT) {
#pragma unroll
                            for (int mo = 0; mo < HT; ++mo) ab[(mi + 1) & 1][mo] = w_d3[((mi + 1) * HT + mo) * 64 + lane];
                        }
                    } else {
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) a[mo] = w_d3[(mi * HT + mo) * 64 + lane];
                    }
                    const int act = mi - start;              // 0 = park; trip k = act - 1: bytes in this slot, table + adds in the next
                    const bool table = pipe && act >= 2;     // trip act - 2: its bytes were read one slot ago
                    float tv[8];
                    if (table) {
                        const int t = (act - 2) * 32;
#pragma unroll
                        for (int k = 0; k < 8; ++k) tv[k] = *reinterpret_cast<const float*>(tb + t * 128 + k * 512 + raw[k] * 4);
                        if (checkN) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) seenN |= (t + 4 * k + g < L) ? (unsigned)lut_s[raw[k]] : 0u;
                        }
                    }
                    if (pipe && act >= 1 && act - 1 < p.Lpad / 32) {
                        const int t = (act - 1) * 32;
#pragma unroll
                        for (int k = 0; k < 8; ++k) raw[k] = rp[t + 4 * k];
                    }
                    if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mi == HT - 1 && r >= p.rlh) break;
#pragma unroll
                        for (int mo = 0; mo < HT; ++mo) h[mo] = mfma16(a[mo][r], h2[mi][r], h[mo]);
                    }
                    if (table) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) sN += tv[k];
                    }
                    if (pipe && act == 0) l1_park();
                }
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) h[mo] = relu4(h[mo]);
                float yy[1];
                f4 hh[HT][1];
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) hh[mo][0] = h[mo];
                final_dot<HT, 1>(db + 4 + 48 * HT, db[4 + 64 * HT], hh, yy, g);
                y = yy[0];
            }
            if (g == 0 && n < p.N) p.out[n * p.out_sn + (p.m_off + m) * p.out_sm] = fx_nan_to_num(y);
            FX_TILE_DONE();
            if (t_nxt >= 0 && !pipe) l1_full(t_nxt);        // the ragged last tile of the batch
            t_cur = t_nxt;
        }
        bad |= seenN >= 0x80u;
    }
    fx_stamp(p.trace, 6);
    if (bad) fx_raise(p.err, FX_ERR_BADCHAR);
}

template <int KIND, int HT, int NLD, int VAR>
int launch_pipe_var(fx_engine* e, const DenseArgs& a, size_t lds_bytes) {
    if (e->rows_req.on) return FX_EUNSUPPORTED;             // (this form asks for the next tile's bytes a tile ahead: not for rows that arrive during the run)
    auto kern = k_score_dense_pipe<KIND, HT, NLD, VAR>;
    static bool attr_set[64] = {};
    if (!attr_set[e->device & 63]) {
        FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[e->device & 63] = true;
    }
    const int64_t U = (int64_t)a.M * a.TG;
    int64_t blocks = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    if (blocks > U) blocks = U;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, e->stream, a);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

template <int KIND, int HT, int NLD>
int launch_pipe(fx_engine* e, const DenseArgs& a, size_t lds_bytes) {
    // dense_pipe: 1 = loads placed by the compiler, 2 = A operands double-buffered by hand (H <= 112: registers)
    if constexpr (HT <= 7) {
        if (e->dense_pipe == 2) return launch_pipe_var<KIND, HT, NLD, 1>(e, a, lds_bytes);
    }
    return launch_pipe_var<KIND, HT, NLD, 0>(e, a, lds_bytes);
}

