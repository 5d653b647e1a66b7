// The layer-parallel form of the protein CNN (k_score_cnn_lp, round 4) and its launcher: explorer-size batches of long sequences layer by
// layer across the chip.  Included by score_cnn_pair.hip inside its anonymous namespace, after PairArgs and the pair kernel whose helpers it
// shares (not a stand-alone header).
#pragma once

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

