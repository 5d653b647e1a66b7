// Dispatch of the fused CNN scoring kernels (template in score_cnn_kernel.h).
#include "score_cnn_kernel.h"

namespace {

// A = 4 (DNA / RNA): NT1 tiles; 16 waves (4 per SIMD) once every CU has plenty of tiles, else 8.
// Hidden sizes other than the canonical 7 tiles get the auto geometry only; HT >= 8 needs the
// 256-register budget of 8-wave workgroups for the dense head.
template <int HT_>
int dispatch_a4(fx_engine* e, const CnnArgs& a0, int variant, bool big, size_t full, size_t conv_only) {
    CnnArgs a = a0;
    const bool dl = full <= (size_t)e->max_lds;          // whole member in LDS, else dense head streams from L2
    const size_t lds = dl ? full : conv_only;
    if (lds > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    if constexpr (HT_ == 7) {
        // the sequences lie in host memory (zero-copy host calls): the forms that copy a tile's bytes into LDS first (STG) -- same
        // walk, same bits -- where the scratch fits beside the weights
        if (e->ascii_host && e->cnn_stage_host && dl && (variant == 0 || variant == 7 || variant == 10 || variant == 11) &&
            lds + 16 * ((16 * (size_t)a.L + 15) / 16 * 16) <= (size_t)e->max_lds) {
            a.TG = (a.N + 15) / 16;
            const int64_t U = (int64_t)a.M * a.TG;
            const bool seg_size = e->cnn_seg != 0 && U <= e->num_cus;       // (explorer-size launches keep their own forms below)
            if (!seg_size) {
                if (big && a.L == 8) return launch_g<4, 5, 2, 7, 1, true, 16, true, 4, true, false, true, true>(e, a, lds);
                if (big && a.L == 14) return launch_g<4, 5, 2, 7, 1, true, 16, true, 10, true, false, true, true>(e, a, lds);
                if (big) return launch_g<4, 5, 2, 7, 1, true, 16, true, 0, false, false, true, true>(e, a, lds);
                if (a.L == 8) return launch_g<4, 5, 2, 7, 1, true, 8, true, 4, true, false, true, true>(e, a, lds);
                return launch_g<4, 5, 2, 7, 1, true, 8, true, 0, false, false, true, true>(e, a, lds);
            }
        }
        // canonical short landscapes get a fully unrolled position loop with s_setprio around the MFMA clusters
        // (+5 % and +2-4 % resp., interleaved A/B: profiles/archive/r1_run9, r1_run16, r1_run18):
        // TF-binding (L = 8) and the RNA landscapes (L = 14)
        if (dl && variant == 0 && big && a.L == 8) variant = 7;
        // ... and in 8-wave workgroups for small and mid-size launches (one or two waves per SIMD: the unrolled walk is
        // 5 % shorter per tile than the ring loop, profiles/archive/r2_trace_probe)
        if (dl && variant == 0 && !big && a.L == 8) variant = 11;
        if (dl && variant == 0 && big && a.L == 14) variant = 10;
        if (dl && variant != 0) {
            const int nt = (variant == 2 || variant == 3) ? 2 : 1;
            a.TG = (a.N + 16 * nt - 1) / (16 * nt);
            switch (variant) {
                case 1: return launch_inst<4, 5, 2, 7, 1, true, 8>(e, a, lds);
#if defined(FX_AB)   // (two tiles per wave, and the unrolled forms without s_setprio: measured losers, A/B build only)
                case 2: return launch_inst<4, 5, 2, 7, 2, true, 4>(e, a, lds);
                case 3: return launch_inst<4, 5, 2, 7, 2, true, 8>(e, a, lds);
                case 5:                                  // variant 4 with the position loop unrolled (TF-binding: L = 8)
                    if (a.L != 8) return fx_fail(e, FX_EINVAL, "cnn_variant 5 is the seq_len = 8 specialisation");
                    return launch_g<4, 5, 2, 7, 1, true, 16, true, 4, false>(e, a, lds);   // no s_setprio (A/B baseline)
                case 6:
                    if (a.L != 14) return fx_fail(e, FX_EINVAL, "cnn_variant 6 is the seq_len = 14 specialisation");
                    return launch_g<4, 5, 2, 7, 1, true, 16, true, 10>(e, a, lds);
#endif
                case 4: return launch_inst<4, 5, 2, 7, 1, true, 16>(e, a, lds);
                case 10:                                 // variant 6 with s_setprio (A/B)
                    if (a.L != 14) return fx_fail(e, FX_EINVAL, "cnn_variant 10 is a seq_len = 14 specialisation");
                    return launch_g<4, 5, 2, 7, 1, true, 16, true, 10, true>(e, a, lds);
                case 7:                                  // unrolled L = 8 specialisation with s_setprio (the default)
                    if (a.L != 8) return fx_fail(e, FX_EINVAL, "cnn_variant 7 is a seq_len = 8 specialisation");
                    // ... with the (tiles mod 4) last tiles of a workgroup walked by wave quads (round 6; cnn_quad_tail = 0: A/B)
                    if (e->cnn_quad_tail && !e->rows_req.on && lds + (size_t)3 * 2 * 8 * 1024 <= (size_t)e->max_lds) {
                        a.quad_tail = 1;
                        return launch_g<4, 5, 2, 7, 1, true, 16, true, 4, true, false, true, false, true>(e, a, lds);
                    }
                    return launch_g<4, 5, 2, 7, 1, true, 16, true, 4, true>(e, a, lds);
                case 11:                                 // unrolled L = 8 form in 8-wave workgroups (256-register budget): small launches
                    if (a.L != 8) return fx_fail(e, FX_EINVAL, "cnn_variant 11 is a seq_len = 8 specialisation");
                    return launch_g<4, 5, 2, 7, 1, true, 8, true, 4, true>(e, a, lds);
                default: return fx_fail(e, FX_EINVAL, "cnn_variant must be 0..11");
            }
        }
    }
    a.TG = (a.N + 15) / 16;
    {
        // small batch of long sequences (an explorer's 1-100 sequence call on an RNA landscape): the 8 waves of a
        // workgroup split one tile's positions instead of 7 of them idling
        const int64_t U = (int64_t)a.M * a.TG;
        const int L1 = a.L - 5 + 1;
        // (one 8-wave workgroup per tile pays from 24 positions; spread over several 4-wave workgroups, from 13 -- shorter
        //  sequences take the quad form)
        const bool multi = HT_ == 7 && dl && e->cnn_seg < 0 && e->cnn_seg_multi && 2 * U <= e->num_cus;
        const bool seg = !e->rows_req.on && e->cnn_seg != 0 && variant == 0 && !e->cnn_conv1_mfma && U <= e->num_cus &&
                         (e->cnn_seg > 0 || L1 >= (multi ? 13 : 24)) && lds + 8 * 2 * 64 * 16 <= (size_t)e->max_lds;
        if (seg) {
            // Long sequences: cut the positions over several workgroups as well (halo: 3 positions per side).  As many
            // segments (>= 2 positions) as one wave of the grid holds, from 4-wave workgroups -- one wave per SIMD -- when
            // that gives as many as 8-wave ones would (score_cnn_pair.hip has the measurements behind both rules).
            a.seg_sb = 1;
            int waves = 8;
            if constexpr (HT_ == 7) {
                if (dl && e->cnn_seg < 0 && e->cnn_seg_multi) {
                    auto count = [&](int w) {
                        int64_t sb = L1 / (w * 2);
                        if (sb > 64 / w) sb = 64 / w;
                        if (sb > e->num_cus / U) sb = e->num_cus / U;
                        return sb < 1 ? (int64_t)1 : sb;
                    };
                    const int64_t sb8 = count(8), sb4 = count(4);
                    if (sb4 * 4 >= sb8 * 8) { waves = 4; a.seg_sb = (int)sb4; }
                    else a.seg_sb = (int)sb8;
                    if (a.seg_sb > 1) {
                        void* ws = nullptr;
                        const size_t pool_bytes = (size_t)U * 2 * 64 * 4 * sizeof(unsigned), cnt_bytes = (size_t)U * sizeof(unsigned);
                        if (int rc = fx_zero_pool(e, pool_bytes + cnt_bytes, &ws)) return rc;
                        a.seg_pool = (unsigned*)ws;
                        a.seg_cnt = (unsigned*)((char*)ws + pool_bytes);
                    }
                    if (waves == 4) return launch_g<4, 5, 2, 7, 1, true, 4, true, 0, false, true>(e, a, lds);
                }
            }
            return dl ? launch_g<4, 5, 2, HT_, 1, true, 8, true, 0, false, true>(e, a, lds)
                      : launch_g<4, 5, 2, HT_, 1, false, 8, true, 0, false, true>(e, a, lds);
        }
    }
    if constexpr (HT_ <= 7) {
        if (big) return dl ? launch_inst<4, 5, 2, HT_, 1, true, 16>(e, a, lds) : launch_inst<4, 5, 2, HT_, 1, false, 16>(e, a, lds);
    }
    return dl ? launch_inst<4, 5, 2, HT_, 1, true, 8>(e, a, lds) : launch_inst<4, 5, 2, HT_, 1, false, 8>(e, a, lds);
}

}  // namespace

int fx_launch_score_cnn_mfma(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                             int64_t N, float* d_out_NM, int Mtot, int m_off) {
    if (N == 0) return FX_OK;
    const FxShape& s = models[0]->shape;
    const FxPackLayout& lay = models[0]->layout;
    for (int m = 0; m < M; ++m) {
        const FxShape& t = models[m]->shape;
        if (t.kind != FX_CNN || t.L != s.L || t.A != s.A || t.F != s.F || t.H != s.H || t.K != s.K)
            return FX_EUNSUPPORTED;                      // heterogeneous: caller scores them one by one
    }
    // num_filters 17..32 (two channel tiles; missing channels are zero padding of the packed blocks); kernel_size 5
    // everywhere, 3 and 7 for the canonical hidden width (97..112 units); everything else -> shape-agnostic kernels
    if ((lay.FT != 2 && !(lay.FT == 1 && s.K == 5 && lay.HT == 7)) || (s.A != 4 && s.A != 20 && s.A != 2)) return FX_EUNSUPPORTED;
    if (s.K != 5 && !((s.K == 3 || s.K == 7) && lay.HT == 7)) return FX_EUNSUPPORTED;
    if (M > FX_MAX_M) return FX_EINVAL;
    // a launched-first host call (rows arrive while the kernel runs): the whole-tile-per-wave forms of the 4-letter CNN only
    if (e->rows_req.on && !(s.A == 4 && s.K == 5 && lay.FT == 2)) return FX_EUNSUPPORTED;
    if (!e->rows_req.on) {
        const int rc = fx_launch_score_cnn_quad(e, models, M, d_ascii, N, d_out_NM, Mtot, m_off);
        if (rc != FX_EUNSUPPORTED) return rc;
    }

    CnnArgs a{};
    a.ascii = d_ascii; a.lut = e->d_lut; a.out = d_out_NM; a.err = e->d_err;
    if (int rc = fx_trace_buffer(e, &a.trace)) return rc;
    a.wave_prio = (int)e->wave_prio;
    a.stage_fill = (int)e->stage_fill;
    for (int m = 0; m < M; ++m) a.w[m] = models[m]->d_packed;
    a.out_sn = e->planar_stride ? 1 : Mtot; a.out_sm = e->planar_stride ? e->planar_stride : 1;
    a.N = N; a.M = M; a.Mtot = Mtot; a.m_off = m_off; a.L = s.L; a.rlh = (lay.HTR == lay.HT) ? lay.RLH : 4;
    a.off_first = (int)lay.off_first; a.off_c2 = (int)lay.off_c2; a.off_c3 = (int)lay.off_c3;
    a.off_cb = (int)lay.off_cb; a.off_w1p = (int)lay.off_w1p; a.conv_floats = (int)lay.conv_floats; a.off_d1 = (int)lay.off_d1;
    a.off_d2 = (int)lay.off_d2; a.off_db = (int)lay.off_db; a.total_floats = (int)lay.total_floats;

#if defined(FX_AB)
    // mean-only call with the rows resident in device memory and every member in this launch: the last member to finish a tile averages
    // it (fx_fused_mean_tile) -- no mean kernel behind the launch.  (Rows that arrive while the kernel runs keep the mean kernel: a
    // starved launch is redone, and its tickets would be left half drawn.)
    if (e->fuse_mean_batch_out && e->planar_stride && m_off == 0 && M == Mtot && M > 1 && M < 8 && !e->rows_req.on) {
        void* ws = nullptr;
        if (int rc = fx_zero_pool(e, (size_t)((N + 15) / 16) * sizeof(unsigned), &ws)) return rc;
        a.fm_mean = e->fuse_mean_batch_out;
        a.fm_cnt = (unsigned*)ws;
    }
#endif
    const size_t full = (size_t)lay.total_floats * 4 + 256 + 48, conv_only = (size_t)lay.conv_floats * 4 + 256 + 48;
    if (s.A == 2) {
        // binary alphabet (`BA = "01"`, sequence_utils.py:16): conv3 has ONE tap (kernel_size = len(alphabet) - 1);
        // canonical filter / hidden / kernel sizes only, first conv in gather form
        if (lay.FT != 2 || lay.HT != 7 || s.K != 5 || full > (size_t)e->max_lds || e->cnn_conv1_mfma) return FX_EUNSUPPORTED;
        a.TG = (N + 15) / 16;
        const bool big = a.TG * M >= (int64_t)e->num_cus * e->cnn_big_units;
        const int L1 = s.L - s.K + 1;
        const bool seg = e->cnn_seg != 0 && (int64_t)M * a.TG <= e->num_cus && (e->cnn_seg > 0 || L1 >= 24) &&
                         full + 8 * 2 * 64 * 16 <= (size_t)e->max_lds;
        if (seg) return launch_g<2, 5, 2, 7, 1, true, 8, true, 0, false, true>(e, a, full);
        return big ? launch_g<2, 5, 2, 7, 1, true, 16, true>(e, a, full) : launch_g<2, 5, 2, 7, 1, true, 8, true>(e, a, full);
    }
    if (lay.FT == 1) {
        // num_filters <= 16: one channel tile (canonical hidden width and kernel size only)
        if (full > (size_t)e->max_lds || e->cnn_conv1_mfma) return FX_EUNSUPPORTED;
        a.TG = (N + 15) / 16;
        if (s.A == 20) return launch_g<20, 5, 1, 7, 1, true, 8, true>(e, a, full);   // 19-tap window of one tile: 8 waves fit
        const bool big = ((N + 15) / 16) * M >= (int64_t)e->num_cus * e->cnn_big_units;
        const int L1 = s.L - s.K + 1;
        const bool seg = e->cnn_seg != 0 && (int64_t)M * a.TG <= e->num_cus && (e->cnn_seg > 0 || L1 >= 24) &&
                         full + 8 * 1 * 64 * 16 <= (size_t)e->max_lds;
        if (seg) return launch_g<4, 5, 1, 7, 1, true, 8, true, 0, false, true>(e, a, full);
        return big ? launch_g<4, 5, 1, 7, 1, true, 16, true>(e, a, full) : launch_g<4, 5, 1, 7, 1, true, 8, true>(e, a, full);
    }
    if (s.A == 4 && s.K != 5) {
        // other kernel sizes: the generic-length kernels only (no unrolled specialisations, conv1 in gather form)
        if (full > (size_t)e->max_lds || e->cnn_conv1_mfma) return FX_EUNSUPPORTED;
        const bool big = ((N + 15) / 16) * M >= (int64_t)e->num_cus * e->cnn_big_units;
        a.TG = (N + 15) / 16;
        const int L1 = s.L - s.K + 1;
        const bool seg = e->cnn_seg != 0 && (int64_t)M * a.TG <= e->num_cus && (e->cnn_seg > 0 || L1 >= 24) &&
                         full + 8 * 2 * 64 * 16 <= (size_t)e->max_lds;
        if (s.K == 3) {
            if (seg) return launch_g<4, 3, 2, 7, 1, true, 8, true, 0, false, true>(e, a, full);
            return big ? launch_g<4, 3, 2, 7, 1, true, 16, true>(e, a, full) : launch_g<4, 3, 2, 7, 1, true, 8, true>(e, a, full);
        }
        return launch_g<4, 7, 2, 7, 1, true, 8, true>(e, a, full);     // 7-tap window: shifting form, 256-register budget
    }
    if (s.A == 4) {
        // more than 128 hidden units at batch size: the two-kernel path, whose head streams the H x H layer through LDS slabs (score_cnn_split.hip)
        if (lay.HT >= 13 && e->cnn_head_slab && !e->rows_req.on && ((N + 15) / 16) * M >= (int64_t)e->num_cus * 16) return FX_EUNSUPPORTED;
        const bool big = ((N + 15) / 16) * M >= (int64_t)e->num_cus * e->cnn_big_units;   // units per CU from which 16-wave workgroups pay
        const int variant = (int)e->cnn_variant;
        switch (lay.HT) {
            case 1: return dispatch_a4<1>(e, a, variant, big, full, conv_only);
            case 2: return dispatch_a4<2>(e, a, variant, big, full, conv_only);
            case 4: return dispatch_a4<4>(e, a, variant, big, full, conv_only);
            case 7: return dispatch_a4<7>(e, a, variant, big, full, conv_only);
            case 8: return dispatch_a4<8>(e, a, variant, big, full, conv_only);
            case 13: return dispatch_a4<13>(e, a, variant, big, full, conv_only);
            case 16: return dispatch_a4<16>(e, a, variant, big, full, conv_only);
            default: return FX_EUNSUPPORTED;
        }
    }
    // A = 20 (proteins): two-waves-per-tile kernel; the single-wave window form (HT = 7 only) is the A/B baseline
    if (e->cnn_pair) {
        const int rc = fx_launch_score_cnn_pair(e, models, M, d_ascii, N, d_out_NM, Mtot, m_off);
        if (rc != FX_EUNSUPPORTED) return rc;
    }
#if defined(FX_AB)
    if (lay.HT != 7 || s.K != 5 || conv_only > (size_t)e->max_lds) return FX_EUNSUPPORTED;
    a.TG = (N + 15) / 16;
    return launch_inst<20, 5, 2, 7, 1, false, 4>(e, a, conv_only);   // 1 wave / SIMD: 512-VGPR budget for the 19-tap window (0.76 vs 0.92 of peak for the pair form)
#else
    return FX_EUNSUPPORTED;                              // (the one-wave window form of the protein CNN lives in the A/B build)
#endif
}
