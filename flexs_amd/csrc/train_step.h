// Part of the training step (train_core.h includes it; not a stand-alone header): fxt_forward_backward -- forward (training mode), MSE, backward and gradient partials of one slice of a mini-batch.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// Forward + backward of one slice.  `slice` rows [slice * R, slice * R + R) of the mini-batch `step`.
// `ws`: the slice's workspace -- the workgroup's LDS on the device when it fits (activations are written by one phase
// and read by the next: an LDS round trip instead of an L2 one; WSAS = 3), else its row of the global arena (WSAS = 1).
// `W`: the member's weights, staged in LDS by the caller when they fit next to the workspace (WAS = 3), else j.w.
template <int WSAS, int WAS, class D = FxtDimsAny, int MODE = 0>
FXT_HD void fxt_forward_backward(const FxtJob& j, const FxtWg& wg, int step, int slice, const uint8_t* ascii,
                                 const uint8_t* lut, const float* labels, typename FxtMem<WSAS>::F ws,
                                 typename FxtMem<WAS>::CF W, typename FxtMem<WSAS>::F split = nullptr) {
    typedef typename FxtMem<WSAS>::F WsF;
    typedef typename FxtMem<WSAS>::CF WsCF;
    typedef typename FxtMem<WSAS>::I WsI;
    typedef typename FxtMem<WSAS>::CI WsCI;
    typedef typename FxtMem<WAS>::CF WCF;
    constexpr bool SWZ = MODE != 0;
    constexpr int LAY = MODE == 3 ? 2 : (MODE != 0 ? 1 : 0);      // how the position-major arrays are indexed (fxt_xi)
    typedef typename FxtMem<WSAS>::CF4 WsCF4;
    typedef typename FxtPick<SWZ, FxtConvA<WsCF>, FxtConvASwz<WsCF>>::T ConvA;
    typedef typename FxtPick<SWZ, FxtConvGradA<WsCF>, FxtConvGradASwz<WsCF>>::T ConvGradA;
    typedef typename FxtPick<SWZ, FxtConvWGradA<WsCF>, FxtConvWGradASwz<WsCF>>::T ConvWGradA;
    typedef typename FxtPick<SWZ, FxtPosMajorB<WsCF>, FxtPosMajorBSwz<WsCF>>::T PosMajorB;
    // (a canonical instantiation rebuilds the description from its constants -- only the sequence length is a run-time value --
    //  so that everything derived from it below is a constant too)
    FxtNet n_ = D::fixed ? fxt_net(D::kind, D::L > 0 ? D::L : j.net.L, D::A, D::F, D::H, D::K) : j.net;
    if (D::fixed && SWZ) n_.ldx = n_.F;        // (rotated rows are F floats apart)
    const FxtNet n = n_;
    const int R = (D::fixed && D::R > 0) ? D::R : j.R, L = n.L, A = n.A, F = n.F;      // (R_ = 0: the rows per slice stay a run-time value too)
    const FxtWs w = fxt_ws(n, R, MODE >= 2);
    // MODE 2 / 3: the conv kernels' staging buffer behind the workspace, j.split_off taps at a time (the host sized it: a tap is
    // F rows of fxt_ld_w(F) floats in MODE 2, 32 rotated rows of 32 floats in MODE 3)
    [[maybe_unused]] WsF wbuf = ws + w.total;
    [[maybe_unused]] const int stage_taps = j.split_off;
    [[maybe_unused]] const int tap_floats = MODE == 3 ? F * F : F * fxt_ld_w(F);
    [[maybe_unused]] FxtTapRegs<WAS> tap;                  // MODE 3: the tap group in flight (see FxtTapRegs)
    tap.taps = 0;
    [[maybe_unused]] const int G3 = fxt_conv32_group(stage_taps, wg.nthr);
    const FxtLay y = fxt_lay(n, WAS == 3);      // (the LDS image of the weights has padded conv-kernel rows)
    const int ldF = w.ldF, ldw = y.ldw;
    WsI codes = (WsI)(ws + w.codes);
    float* part = j.partial + (long long)slice * fxt_pstride(j);
    const int32_t* order = j.order + (long long)step * j.batch;
    const int slot0 = slice * R;
    const int nwv = wg.nthr >> 6;
    const bool can_split = split != nullptr;
    const int sidx = step % j.steps_per_epoch;
    const int nvalid = (j.n - sidx * j.batch) < j.batch ? (j.n - sidx * j.batch) : j.batch;
    FXT_STAMP(0);
    const float keep_scale = 1.f / (1.f - FXT_DROPOUT);
    const FxtDiv dL1 = fxt_div(n.kind == 0 ? n.L1 : 1), dF = fxt_div(n.kind == 0 ? F : 1), dA = fxt_div(A);

    if constexpr (MODE == 3) {
        if (n.kind == 0) {
            tap.template fetch<0>(wg, W + y.cw[1], 0, G3 < n.K ? G3 : n.K);      // conv2's first tap group: two phases ahead
            FXT_FOR(i, w.ldF, wg) ws[w.zero + i] = 0.f;              // the row of zeros (FxtConvAZ)
        }
    }
    // ---- the slice's rows as alphabet indices (padding slots read row 0: their gradient is zeroed at the loss)
    FXT_FOR(i, R * L, wg) {
        const int r = i / L, l = i - r * L;
        const int slot = slot0 + r;
        const int row = (slot < j.batch && order[slot] >= 0) ? order[slot] : 0;
        codes[i] = lut[ascii[(long long)row * L + l]];
    }
    FXT_FOR(r, R, wg) {                     // (the label fetch is a dependent global load too: issued here, used after the forward)
        const int slot = slot0 + r;
        const bool valid = slot < j.batch && order[slot] >= 0;
        ws[w.ylab + r] = labels[valid ? order[slot] : 0];
        ws[w.yvalid + r] = valid ? 1.f : 0.f;
    }
    // MODE 2: conv1's kernel and bias (K A rows of F floats + F: contiguous in Keras order) into the staging buffer as well, under the
    // same barrier -- conv1 is K gathered kernel rows per output, from L2 otherwise (~7 dependent round trips per thread at one row per slice)
    [[maybe_unused]] bool conv1_staged = false;
    if constexpr (MODE >= 2) {
        if (n.kind == 0 && (n.K * A + 1) * F <= stage_taps * tap_floats) {
            conv1_staged = true;
            FXT_FOR(i, (n.K * A + 1) * F, wg) wbuf[i] = W[y.cw[0] + i];
        }
    }
    fxt_sync_ws<WSAS>(); FXT_STAMP(1);

    WsCF feat = nullptr;            // input of the dense stack when it is not the one-hot
    if (n.kind == 0) {
        const int L1 = n.L1, K = n.K;
        WsF a1 = ws + w.a[0]; WsF a2 = ws + w.a[1]; WsF a3 = ws + w.a[2];
        // conv1 ('valid') on a one-hot input: a sum of K kernel rows
        if constexpr (MODE >= 2) {
            if (conv1_staged) {
                FXT_FOR(i, R * L1 * F, wg) {
                    const int o = i % F, rt = i / F, t = rt % L1, r = rt / L1;
                    float s = wbuf[K * A * F + o];
                    for (int jj = 0; jj < K; ++jj) s += wbuf[(jj * A + codes[r * L + t + jj]) * F + o];
                    a1[fxt_xi<LAY>(rt, o, ldF)] = s > 0.f ? s : 0.f;
                }
            }
        }
        if (MODE < 2 || !conv1_staged)
        FXT_FOR(i, R * L1 * F, wg) {
            const int o = i % F, rt = i / F, t = rt % L1, r = rt / L1;
            float s = W[y.cb[0] + o];
            for (int jj = 0; jj < K; ++jj) s += W[y.cw[0] + (jj * A + codes[r * L + t + jj]) * ldw + o];
            a1[fxt_xi<LAY>(rt, o, ldF)] = s > 0.f ? s : 0.f;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(2);
        {   // conv2 ('same', K taps)
            WCF b = W + y.cb[1];
            struct Put { WsF y; WCF b; int ld; FXT_HD void put(int m, int nn, float v) const { v += b[nn]; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            struct Put3 { WsF y; WCF b; int ld; FXT_HD float pre(int nn) const { return b[nn]; }
                          FXT_HD void put(int m, int nn, float v, float bias) const { v += bias; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            if constexpr (MODE == 3)
                fxt_conv32_staged<WSAS, WAS, 0>(wg, R * L1, K, FxtConvAZ<WsCF, WsCF4>{a1, L1, (K - 1) / 2, w.zero - w.a[0], dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, Put3{a2, b, ldF}, W + y.cw[1], wbuf, G3, tap, true, false, W + y.cw[2], n.K3);
            else if constexpr (MODE == 2)
                fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K, F, ConvA{a1, L1, ldF, (K - 1) / 2, dL1}, FxtConvW<WsCF>{wbuf, F, fxt_ld_w(F)}, Put{a2, b, ldF},
                                      W + y.cw[1], wbuf, stage_taps, F, F, fxt_ld_w(F));
            else
            fxt_gemm(wg, R * L1, F, K, F, ConvA{a1, L1, ldF, (K - 1) / 2, dL1}, FxtConvW<WCF>{W + y.cw[1], F, ldw}, Put{a2, b, ldF}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(3);
        {   // conv3 ('same', A - 1 taps)
            WCF b = W + y.cb[2];
            struct Put { WsF y; WCF b; int ld; FXT_HD void put(int m, int nn, float v) const { v += b[nn]; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            struct Put3 { WsF y; WCF b; int ld; FXT_HD float pre(int nn) const { return b[nn]; }
                          FXT_HD void put(int m, int nn, float v, float bias) const { v += bias; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            if constexpr (MODE == 3)
                fxt_conv32_staged<WSAS, WAS, 0>(wg, R * L1, n.K3, FxtConvAZ<WsCF, WsCF4>{a2, L1, (n.K3 - 1) / 2, w.zero - w.a[1], dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, Put3{a3, b, ldF}, W + y.cw[2], wbuf, G3, tap, true, false, (WCF) nullptr, 0,
                                                [&](int k) {           // profiling aid (train_trace): group 0 of conv3's forward -- every wave's clock at the end of its MFMAs (44 + wave); wave 0's at the group's start (60) and at the next group's (61)
#if FXT_DEVICE
                                                    if (j.dbg && slice == 0 && (wg.tid & 63) == 0) {
                                                        if (k == 1) j.dbg[44 + (wg.tid >> 6)] = wall_clock64();
                                                        if (k == 0 && wg.tid == 0) j.dbg[60] = wall_clock64();
                                                        if (k == 2 && wg.tid == 0) j.dbg[61] = wall_clock64();
                                                    }
#endif
                                                    (void)k; });
            else if constexpr (MODE == 2)
                fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, n.K3, F, ConvA{a2, L1, ldF, (n.K3 - 1) / 2, dL1}, FxtConvW<WsCF>{wbuf, F, fxt_ld_w(F)}, Put{a3, b, ldF},
                                      W + y.cw[2], wbuf, stage_taps, F, F, fxt_ld_w(F));
            else
            fxt_gemm(wg, R * L1, F, n.K3, F, ConvA{a2, L1, ldF, (n.K3 - 1) / 2, dL1}, FxtConvW<WCF>{W + y.cw[2], F, ldw}, Put{a3, b, ldF}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(4);
        WsF g = ws + w.g; WsF cnt = ws + w.cnt;
        bool pooled = false;
        if constexpr (MODE != 0) {
            // The long sequences these modes serve run ONE row per slice: a thread per (row, channel) leaves 32 of 1024 threads with two
            // dependent walks over 233 positions (~20 us of a step).  Here every (row, channel) is shared by FXT_POOL_PARTS threads,
            // position t to thread t mod PARTS; partial maxima and tie counts meet in the (still unused) gradient array dzB.  A maximum
            // and an integer count do not depend on the order they are taken in; the one order-dependent case of the walk below -- a NaN
            // in position 0 stays, a NaN elsewhere is skipped -- is kept: the same bits.
            constexpr int PARTS = 32;
            if (L1 >= 2 * PARTS) {
                pooled = true;
                WsF pmax = ws + w.dzB; WsF pcnt = pmax + R * F * PARTS;        // (2 R F PARTS <= R L1 F floats)
                FXT_FOR(i, R * F * PARTS, wg) {
                    const int part = i % PARTS, rf = i / PARTS, r = rf / F, f = rf - r * F;
                    float mx = -INFINITY;
                    for (int t = part; t < L1; t += PARTS) { const float v = a3[fxt_xi<LAY>(r * L1 + t, f, ldF)]; mx = v > mx ? v : mx; }
                    pmax[i] = mx;
                }
                fxt_sync_ws<WSAS>();
                FXT_FOR(i, R * F * PARTS, wg) {
                    const int part = i % PARTS, rf = i / PARTS, r = rf / F, f = rf - r * F;
                    float mx = pmax[rf * PARTS];
                    for (int q = 1; q < PARTS; ++q) { const float v = pmax[rf * PARTS + q]; mx = v > mx ? v : mx; }
                    const float first = a3[fxt_xi<LAY>(r * L1, f, ldF)];
                    if (first != first) mx = first;
                    int c = 0;
                    for (int t = part; t < L1; t += PARTS) c += a3[fxt_xi<LAY>(r * L1 + t, f, ldF)] == mx;
                    pcnt[i] = (float)c;
                    if (part == 0) g[r * ldF + f] = mx;
                }
                fxt_sync_ws<WSAS>();
                FXT_FOR(i, R * F, wg) {
                    const int r = i / F, f = i - r * F;
                    float c = 0.f;
                    for (int q = 0; q < PARTS; ++q) c += pcnt[i * PARTS + q];      // (small integers: exact in any order)
                    cnt[r * ldF + f] = c;
                }
            }
        }
        if (!pooled)
        FXT_FOR(i, R * F, wg) {             // GlobalMaxPooling1D + the number of positions that attain the maximum
            const int r = i / F, f = i - r * F;
            float mx = a3[fxt_xi<LAY>(r * L1, f, ldF)];
            for (int t = 1; t < L1; ++t) { const float v = a3[fxt_xi<LAY>(r * L1 + t, f, ldF)]; mx = v > mx ? v : mx; }
            int c = 0;
            for (int t = 0; t < L1; ++t) c += a3[fxt_xi<LAY>(r * L1 + t, f, ldF)] == mx;
            g[r * ldF + f] = mx; cnt[r * ldF + f] = (float)c;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(5);
        feat = g;
    }

    // ---- dense stack, forward  (unrolled over the at most four layers: with a compile-time layer index the workspace
    // offsets are registers -- indexed at run time the little offset table lived in scratch memory, a global round trip
    // per access in the middle of the step)
#if FXT_DEVICE
#pragma unroll
#endif
    for (int li = 0; li < FXT_MAX_LAYERS; ++li) {
        if (li >= n.nl) break;
        const int Kd = n.dim[li], Nd = n.dim[li + 1];
        WCF Wl = W + y.w[li];
        WCF bl = W + y.b[li];
        WsF out = ws + w.act[li];
        const int ld_in = (li == 0 && !n.onehot_in) ? ldF : Kd;     // row stride of the layer's input
        const bool last = li == n.nl - 1;
        const bool drop = li == n.drop_layer;
        if (li == 0 && n.onehot_in) {
            FXT_FOR(i, R * Nd, wg) {        // one-hot input: sum of L rows
                const int r = i / Nd, o = i - r * Nd;
                float s = bl[o];
                for (int l = 0; l < L; ++l) s += Wl[(l * A + codes[r * L + l]) * Nd + o];
                out[i] = (last || s > 0.f) ? s : 0.f;
            }
        } else {
            WsCF in = li == 0 ? feat : ws + w.act[li - 1];
            struct Put {
                WsF y; WCF b; int Nd; bool last, drop; const FxtJob* j; int step, slot0; float ks;
                FXT_HD void put(int m, int nn, float v) const {
                    v += b[nn];
                    if (!last) v = v > 0.f ? v : 0.f;
                    if (drop) v = fxt_keep(*j, step, slot0 + m, nn) ? v * ks : 0.f;
                    y[m * Nd + nn] = v;
                }
            };
            fxt_gemm(wg, R, Nd, 1, Kd, FxtRowMajorA<WsCF>{in, ld_in}, FxtRowMajorB<WCF>{Wl, Nd}, Put{out, bl, Nd, last, drop, &j, step, slot0, keep_scale}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(20 + li);
    }

    // ---- loss: d(mean over valid rows of (pred - y)^2) / d pred
    {
        // (selects, not w.act[n.nl - 1]: a run-time index would put the offset table into scratch memory)
        const int last_act = n.nl == 4 ? w.act[3] : (n.nl == 3 ? w.act[2] : (n.nl == 2 ? w.act[1] : w.act[0]));
        const int last_du = n.nl == 4 ? w.du[3] : (n.nl == 3 ? w.du[2] : (n.nl == 2 ? w.du[1] : w.du[0]));
        WsCF pred = ws + last_act;
        WsF du = ws + last_du;
        FXT_FOR(r, R, wg) {
            const float e = ws[w.yvalid + r] != 0.f ? pred[r] - ws[w.ylab + r] : 0.f;
            du[r] = 2.f * e / (float)nvalid;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(7);
        FXT_FOR(i, 1, wg) {                 // the slice's sum of squared errors (fixed order)
            float sse = 0.f;
            for (int r = 0; r < R; ++r) { const float e = du[r] * (float)nvalid * 0.5f; sse += e * e; }
#if FXT_DEVICE
            if (j.agent_io) fxt_store_agent(&part[n.P], sse); else
#endif
            part[n.P] = sse;
        }
    }

    // ---- dense stack, backward
#if FXT_DEVICE
#pragma unroll
#endif
    for (int lq = 0; lq < FXT_MAX_LAYERS; ++lq) {
        const int li = FXT_MAX_LAYERS - 1 - lq;
        if (li >= n.nl) continue;
        const int Kd = n.dim[li], Nd = n.dim[li + 1];
        WCF Wl = W + y.w[li];
        WsCF du = ws + w.du[li];
        const int ld_in = (li == 0 && !n.onehot_in) ? ldF : Kd;     // row stride of the layer's input
        struct PutW {
            float* gw; float* gb; int Kd, Nd; bool agent;
            FXT_HD void put(int m, int nn, float v) const {
                float* p = m < Kd ? gw + m * Nd + nn : gb + nn;
#if FXT_DEVICE
                if (agent) { fxt_store_agent(p, v); return; }
#endif
                *p = v;
            }
        };
        const PutW putw{part + n.off_w[li], part + n.off_b[li], Kd, Nd, j.agent_io != 0};
        if (li == 0 && n.onehot_in) {
            // (one thread per output element -- 8 FMAs each, no per-tile bookkeeping -- was measured SLOWER than the 49
            // two-k-step MFMA tiles here: 7.4 vs 6.1 us for the 100 x 100 layer, profiles/r3_train_trace.log)
#if FXT_DEVICE
            if (Kd + 1 > FXT_ONEHOT_WGRAD_DIRECT && R <= 16) {
                // Round 6, long one-hot inputs (a protein MLP: 1 801 x 200 elements per slice): as an MFMA product this is 1 469 tiles of TWO
                // k-steps each -- ~2.5 us of per-tile bookkeeping for 2 MFMAs, 230 of the step's 257 us (profiles/r6_train_protein_survey.log).
                // Directly instead: a thread owns an output column and a run of positions, keeps the column's R gradient values in registers
                // and, per position, adds to each letter's element the values of the rows that carry the letter, in row order -- the terms
                // the product adds (its other terms are exact zeros; the host build's chain of fmaf over the rows gives the same bits).
                // (Four columns per thread and 16-byte stores measured the same 205 us per step -- the phase is bound by its 1.44 MB of partial
                //  stores per slice -- and that instantiation made CNN cases of the device step fail although no CNN path reads this code:
                //  csrc/OPTIONS.md; one column per thread it is.)
                auto direct = [&](auto rows) {
                    constexpr int RR = decltype(rows)::value;              // 8 (the default slice) or 16 rows, unrolled
                    const int chunks = wg.nthr / Nd > 0 ? wg.nthr / Nd : 1;
                    const int lpc = (L + chunks - 1) / chunks;
                    FXT_FOR(t, chunks * Nd, wg) {
                        const int ch = t / Nd, o = t - ch * Nd;
                        const int l_hi = (ch + 1) * lpc < L ? (ch + 1) * lpc : L;
                        float d[RR];
#pragma unroll
                        for (int r = 0; r < RR; ++r) d[r] = r < R ? du[r * Nd + o] : 0.f;
                        for (int l = ch * lpc; l < l_hi; ++l) {
                            int cr[RR];
#pragma unroll
                            for (int r = 0; r < RR; ++r) cr[r] = r < R ? codes[r * L + l] : -1;
                            for (int c = 0; c < A; ++c) {
                                float v = 0.f;
#pragma unroll
                                for (int r = 0; r < RR; ++r)
                                    if (cr[r] == c) v += d[r];
                                putw.put(l * A + c, o, v);
                            }
                        }
                        if (ch == 0) {
                            float v = 0.f;
#pragma unroll
                            for (int r = 0; r < RR; ++r)
                                if (r < R) v += d[r];
                            putw.put(Kd, o, v);
                        }
                    }
                };
                if (R <= 8) direct(std::integral_constant<int, 8>{});
                else direct(std::integral_constant<int, 16>{});
            } else
#endif
            fxt_gemm(wg, Kd + 1, Nd, 1, R, FxtOneHotWGradA<WsCI>{codes, L, A, Kd, 0, dA}, FxtRowMajorB<WsCF>{du, Nd}, putw);
        } else {
            WsCF in = li == 0 ? feat : ws + w.act[li - 1];
            // gradient w.r.t. the layer's input FIRST (few tiles, Nd k-steps each); through the previous layer's ReLU (and
            // Dropout: a dropped unit's stored output is 0, a kept one carries the 1 / (1 - rate) scale) ...
            if (li > 0) {
                const bool dropped = (li - 1) == n.drop_layer;
                struct PutX { WsF d; WsCF y; int Kd; float ks; FXT_HD void put(int m, int nn, float v) const { d[m * Kd + nn] = y[m * Kd + nn] > 0.f ? v * ks : 0.f; } };
                fxt_gemm(wg, R, Kd, 1, Nd, FxtRowMajorA<WsCF>{du, Nd}, FxtTransB<WCF>{Wl, Nd}, PutX{ws + w.du[li - 1], in, Kd, dropped ? keep_scale : 1.f}, 0, split);
            } else {
                struct PutG { WsF d; int ld; FXT_HD void put(int m, int nn, float v) const { d[m * ld + nn] = v; } };
                fxt_gemm(wg, R, Kd, 1, Nd, FxtRowMajorA<WsCF>{du, Nd}, FxtTransB<WCF>{Wl, Nd}, PutG{ws + w.dg, ld_in}, 0, split);
            }
            // ... then the weight gradient (many tiles of R / 4 k-steps), dealt on from the wave behind the last long tile
            fxt_gemm(wg, Kd + 1, Nd, 1, R, FxtDenseWGradA<WsCF>{in, Kd, ld_in}, FxtRowMajorB<WsCF>{du, Nd}, putw, fxt_jobs(R, Kd, 1, Nd, nwv, can_split));
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(30 + li);
    }

    if (n.kind == 0) {
        const int L1 = n.L1, K = n.K, K3 = n.K3;
        WsCF a1 = ws + w.a[0]; WsCF a2 = ws + w.a[1]; WsCF a3 = ws + w.a[2];
        WsCF g = ws + w.g; WsCF cnt = ws + w.cnt; WsCF dg = ws + w.dg;
        WsF dzA = ws + w.dzA; WsF dzB = ws + w.dzB;
        if constexpr (MODE == 3) tap.template fetch<1>(wg, W + y.cw[2], 0, G3 < K3 ? G3 : K3);     // conv3's first group for the input gradient, behind this phase
        FXT_FOR(i, R * L1 * F, wg) {        // max-pool backward (ties share evenly) through conv3's ReLU
            const int f = i % F, rt = i / F, r = rt / L1;
            const float v = a3[fxt_xi<LAY>(rt, f, ldF)];
            dzA[fxt_xi<LAY>(rt, f, ldF)] = (v > 0.f && v == g[r * ldF + f]) ? dg[r * ldF + f] / cnt[r * ldF + f] : 0.f;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(9);
        struct PutW {
            float* gw; float* gb; int rows, F; bool agent;
            FXT_HD void put(int m, int nn, float v) const {
                float* p = m < rows ? gw + m * F + nn : gb + nn;
#if FXT_DEVICE
                if (agent) { fxt_store_agent(p, v); return; }
#endif
                *p = v;
            }
        };
        const bool ag = j.agent_io != 0;
        struct PutX { WsF d; WsCF y; int ld; FXT_HD void put(int m, int nn, float v) const { d[fxt_xi<LAY>(m, nn, ld)] = y[fxt_xi<LAY>(m, nn, ld)] > 0.f ? v : 0.f; } };
        struct PutX3 { WsF d; WsCF y; int ld; FXT_HD int pre(int) const { return 0; }
                       FXT_HD void put(int m, int nn, float v, int) const { d[fxt_xi<LAY>(m, nn, ld)] = y[fxt_xi<LAY>(m, nn, ld)] > 0.f ? v : 0.f; } };
        // conv3: input gradient (few tiles, K3 x F / 4 k-steps) first, the weight gradient dealt on behind it
        if constexpr (MODE == 3)
            fxt_conv32_staged<WSAS, WAS, 1>(wg, R * L1, K3, FxtConvGradAZ<WsCF, WsCF4>{dzA, L1, (K3 - 1) / 2, w.zero - w.dzA, dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, PutX3{dzB, a2, ldF}, W + y.cw[2], wbuf, G3, tap, true, false,
                                            W + y.cw[1], K);          // (leaves conv2's first group in `tap`: fetched BEFORE this phase's partial stores)
        else if constexpr (MODE == 2)
            fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K3, F, ConvGradA{dzA, L1, ldF, (K3 - 1) / 2, dL1}, FxtConvGradW<WsCF>{wbuf, F, fxt_ld_w(F)}, PutX{dzB, a2, ldF},
                                  W + y.cw[2], wbuf, stage_taps, F, F, fxt_ld_w(F));
        else
        fxt_gemm(wg, R * L1, F, K3, F, ConvGradA{dzA, L1, ldF, (K3 - 1) / 2, dL1}, FxtConvGradW<WCF>{W + y.cw[2], F, ldw}, PutX{dzB, a2, ldF}, 0, split);
        if constexpr (MODE == 3)
        {   FXT_STAMP(40);
            // the hook: conv2's group goes into the staging buffer (every wave is through with conv3's last group behind the barrier)
            // before this phase's 78 KiB of partial stores are issued; conv2's input gradient then starts without touching global memory
            auto commit = [&]() { fxt_sync_ws<WSAS>(); tap.template store<WSAS>(wg, wbuf); FXT_STAMP(41); };
            fxt_conv32_wgrad<D::fixed ? D::A - 1 : 0>(wg, R, L1, K3, (K3 - 1) / 2, a2, (WsCF)dzA, (WsCF)(ws + w.zero), PutW{part + n.off_cw[2], part + n.off_cb[2], K3 * F, F, ag}, commit);
        }
        else
        fxt_gemm(wg, K3 * F + 1, F, R, L1, ConvWGradA{a2, L1, F, ldF, (K3 - 1) / 2, K3 * F, dF}, PosMajorB{dzA, L1, ldF}, PutW{part + n.off_cw[2], part + n.off_cb[2], K3 * F, F, ag}, fxt_jobs(R * L1, F, K3, F, nwv, can_split));
        fxt_sync_ws<WSAS>(); FXT_STAMP(10);
        // conv2
        if constexpr (MODE == 3)
            fxt_conv32_staged<WSAS, WAS, 1>(wg, R * L1, K, FxtConvGradAZ<WsCF, WsCF4>{dzB, L1, (K - 1) / 2, w.zero - w.dzB, dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, PutX3{dzA, a1, ldF}, W + y.cw[1], wbuf, G3, tap, false, true);
        else if constexpr (MODE == 2)
            fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K, F, ConvGradA{dzB, L1, ldF, (K - 1) / 2, dL1}, FxtConvGradW<WsCF>{wbuf, F, fxt_ld_w(F)}, PutX{dzA, a1, ldF},
                                  W + y.cw[1], wbuf, stage_taps, F, F, fxt_ld_w(F));
        else
        fxt_gemm(wg, R * L1, F, K, F, ConvGradA{dzB, L1, ldF, (K - 1) / 2, dL1}, FxtConvGradW<WCF>{W + y.cw[1], F, ldw}, PutX{dzA, a1, ldF}, 0, split);
        if constexpr (MODE == 3)
        {   FXT_STAMP(42);
            fxt_conv32_wgrad<D::fixed ? D::K : 0>(wg, R, L1, K, (K - 1) / 2, a1, (WsCF)dzB, (WsCF)(ws + w.zero), PutW{part + n.off_cw[1], part + n.off_cb[1], K * F, F, ag});
            FXT_STAMP(43);
        }
        else
        fxt_gemm(wg, K * F + 1, F, R, L1, ConvWGradA{a1, L1, F, ldF, (K - 1) / 2, K * F, dF}, PosMajorB{dzB, L1, ldF}, PutW{part + n.off_cw[1], part + n.off_cb[1], K * F, F, ag}, fxt_jobs(R * L1, F, K, F, nwv, can_split));
        fxt_sync_ws<WSAS>(); FXT_STAMP(11);
        // conv1 (one-hot input, 'valid')
        if constexpr (MODE == 3)
            fxt_gemm(wg, K * A + 1, F, R, L1, FxtOneHotWGradA<WsCI>{codes, L, A, K * A, 1, dA}, FxtPosMajorBL<WsCF, 2>{dzA, L1, ldF}, PutW{part + n.off_cw[0], part + n.off_cb[0], K * A, F, ag});
        else
        fxt_gemm(wg, K * A + 1, F, R, L1, FxtOneHotWGradA<WsCI>{codes, L, A, K * A, 1, dA}, PosMajorB{dzA, L1, ldF}, PutW{part + n.off_cw[0], part + n.off_cb[0], K * A, F, ag});
    }
    FXT_STAMP(63);
}

