// ASAN / UBSAN exercise of the host-compilable sources the device kernels are built from (test infrastructure):
//   csrc/train_core.h  one training step of every architecture at odd shapes (index functors, workspace layout,
//                      magic-number division, the fixed-order Adam reduction), threads as loops;
//   csrc/myers.h       the bit-parallel Levenshtein in its register form and in strips, against a plain DP.
// Built and run by tests/test_sanitizers.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <algorithm>
#include <array>
#include "../../flexs_amd/csrc/myers.h"
#include "../../flexs_amd/csrc/host_collect.cc"      // fx_collect_lines: the vector path of the resident form's answer collection
#include <cstdlib>
#include "../../flexs_amd/csrc/train_core.h"

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

static int dp(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b) {
    std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
    for (size_t j = 0; j <= b.size(); ++j) prev[j] = (int)j;
    for (size_t i = 1; i <= a.size(); ++i) {
        cur[0] = (int)i;
        for (size_t j = 1; j <= b.size(); ++j) {
            const int sub = prev[j - 1] + (a[i - 1] != b[j - 1]);
            cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), sub);
        }
        std::swap(prev, cur);
    }
    return prev[b.size()];
}

template <int LW>
static int strips(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b) {
    std::vector<signed char> h(b.size() + 1, 0);
    const int la = (int)a.size(), lb = (int)b.size();
    const int ns = la > 0 ? (la + 64 * LW - 1) / (64 * LW) : 1;
    int part = 0;
    for (int s = 0; s < ns; ++s) {
        const int r0 = s * 64 * LW, rows = std::min(la - r0, 64 * LW);
        std::vector<uint64_t> peq((size_t)256 * LW, 0);
        for (int i = 0; i < rows; ++i) peq[(size_t)a[r0 + i] * LW + (i >> 6)] |= 1ull << (i & 63);
        part = fx_myers_strip<LW>(rows > 0 ? rows : 0, lb, [&](int c, int w) { return peq[(size_t)c * LW + w]; },
                                  [&](int i) { return (int)b[i]; }, h.data(), h.data(), 1, s == 0, s == ns - 1);
    }
    return la + part;
}

// One mini-batch step.  plain_rows: the workspace's activation rows unpadded (FxtNet::ldx = F -- what the host falls back to when the
// padded workspace misses the LDS budget); out_w: the updated weights (the two layouts must give the same bits).
// rotated: plain rows stored rotated (train_core.h "Rotated rows", F a power of two) -- again the same bits.
// stage_taps > 0 (with rotated): MODE 2 -- the gradient array over the last conv output and the conv kernels staged through a buffer
// behind the workspace, that many taps at a time -- and again the same bits.
static int train_case(int kind, int L, int A, int F, int H, int K, int rows, int R, bool plain_rows = false, std::vector<float>* out_w = nullptr,
                      unsigned seed = 0, bool rotated = false, int stage_taps = 0, bool c32 = false) {
    if (seed) rng_state = seed;
    FxtJob j{};
    j.net = fxt_net(kind, L, A, kind == 0 ? F : 0, H, kind == 0 ? K : 0);
    if (plain_rows && kind == 0) j.net.ldx = j.net.F;
    j.batch = rows; j.steps_per_epoch = 1; j.total_steps = 1; j.n = rows; j.R = R; j.S = (rows + R - 1) / R;
    // exact-size heap buffers: any index past an array is an ASAN report
    std::vector<float> w((size_t)j.net.P), m((size_t)j.net.P, 0.f), v((size_t)j.net.P, 0.f), partial((size_t)j.S * (j.net.P + 1), 0.f);
    for (auto& x : w) x = ((int)(rnd() % 2001) - 1000) * 1e-4f;
    std::vector<int32_t> order((size_t)rows);
    for (int i = 0; i < rows; ++i) order[(size_t)i] = (rnd() % 7 == 0) ? -1 : (int)(rnd() % rows);      // some padding slots
    std::vector<uint8_t> ascii((size_t)rows * L), lut(256, 0xFF), keep((size_t)rows * H);
    for (int a = 0; a < A; ++a) lut[65 + a] = (uint8_t)a;
    for (auto& c : ascii) c = (uint8_t)(65 + rnd() % A);
    for (auto& k : keep) k = rnd() % 4 != 0;
    std::vector<float> labels((size_t)rows);
    for (auto& y : labels) y = (rnd() % 1000) * 1e-3f;
    const float lr = 1e-3f;
    float loss = 0.f;
    j.w = w.data(); j.adam_m = m.data(); j.adam_v = v.data(); j.partial = partial.data(); j.order = order.data();
    j.keep = kind == 0 ? keep.data() : nullptr; j.lr_t = &lr; j.step_loss = &loss;
    j.ws_slice = fxt_ws(j.net, R).total;
    if (stage_taps > 0) {
        j.ws_slice = fxt_ws(j.net, R, true).total + stage_taps * j.net.F * fxt_ld_w(j.net.F);      // exact size: four arrays + the tap buffer
        j.split_off = stage_taps;
    }
    if (c32) j.ws_slice = fxt_ws(j.net, R, true).total + stage_taps * 1024;      // MODE 3: a tap = 32 rotated rows of 32 floats
    std::vector<float> ws((size_t)j.S * (size_t)j.ws_slice, 0.f);
    j.ws = ws.data();
    for (int s = 0; s < j.S; ++s) {
        if (c32) fxt_forward_backward<0, 0, FxtDimsAny, 3>(j, FxtWg{0, 1}, 0, s, ascii.data(), lut.data(), labels.data(), j.ws + (long long)s * j.ws_slice, (const float*)j.w);
        else if (rotated && stage_taps > 0) fxt_forward_backward<0, 0, FxtDimsAny, 2>(j, FxtWg{0, 1}, 0, s, ascii.data(), lut.data(), labels.data(), j.ws + (long long)s * j.ws_slice, (const float*)j.w);
        else if (rotated) fxt_forward_backward<0, 0, FxtDimsAny, 1>(j, FxtWg{0, 1}, 0, s, ascii.data(), lut.data(), labels.data(), j.ws + (long long)s * j.ws_slice, (const float*)j.w);
        else fxt_forward_backward<0, 0>(j, FxtWg{0, 1}, 0, s, ascii.data(), lut.data(), labels.data(), j.ws + (long long)s * j.ws_slice, (const float*)j.w);
    }
    fxt_step_loss(j, 0);
    for (int i = 0; i < j.net.P; ++i) fxt_adam(j, 0, i);
    for (float x : w) if (!(x == x)) { std::printf("NaN weight: kind %d L %d\n", kind, L); return 1; }
    if (out_w) *out_w = w;
    return 0;
}

int main() {
    int bad = 0;
    const int shapes[][8] = {  // kind, L, A, F, H, K, rows, R
        {0, 8, 4, 32, 100, 5, 40, 8}, {0, 9, 4, 8, 16, 3, 37, 5}, {0, 12, 20, 5, 7, 4, 19, 16}, {0, 6, 2, 3, 5, 2, 11, 1}, {0, 7, 4, 1, 1, 7, 3, 4},
        {1, 14, 4, 0, 100, 0, 48, 16}, {1, 5, 20, 0, 9, 0, 5, 3}, {1, 1, 2, 0, 1, 0, 1, 1}, {2, 30, 20, 0, 100, 0, 33, 8}, {2, 9, 4, 0, 20, 0, 37, 64}};
    for (const auto& s : shapes) bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
    {   // rotated rows on the shapes they are meant for: a 20-letter alphabet, ONE row per slice (long sequences), 32 / 64 / 16 filters
        const int rot[][8] = {{0, 41, 20, 32, 100, 5, 5, 1}, {0, 23, 20, 64, 9, 4, 3, 1}, {0, 19, 20, 16, 12, 3, 9, 2}, {0, 37, 6, 32, 7, 6, 4, 3},
                              {0, 70, 4, 32, 16, 5, 2, 1}, {0, 69, 6, 64, 5, 6, 1, 1}};      // (>= 64 positions: the max-pool shared by 32 threads per channel)
        for (const auto& s : rot) {
            std::vector<float> wa, wc;
            bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &wa, 777u, false);
            bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &wc, 777u, true);
            if (wa.size() != wc.size() || std::memcmp(wa.data(), wc.data(), wa.size() * sizeof(float)) != 0) { std::printf("rotated rows differ: L %d F %d\n", s[1], s[3]); ++bad; }
            if (s[3] & 31) continue;                       // staged conv kernels: whole groups of eight k-steps per tap
            for (int taps : {1, 2, 6, 64}) {               // (64: more than any kernel has -- one group)
                std::vector<float> wd;
                bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &wd, 777u, true, taps);
                if (wa.size() != wd.size() || std::memcmp(wa.data(), wd.data(), wa.size() * sizeof(float)) != 0) { std::printf("staged conv kernels differ: L %d F %d taps %d\n", s[1], s[3], taps); ++bad; }
            }
            if (s[3] != 32) continue;                      // MODE 3 (round 5): the F = 32 form -- rotated kernel rows in the tap buffer, the weight gradient on its own loop
            for (int taps : {1, 3, 8}) {
                std::vector<float> we;
                bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &we, 777u, true, taps, true);
                if (wa.size() != we.size() || std::memcmp(wa.data(), we.data(), wa.size() * sizeof(float)) != 0) { std::printf("F = 32 form differs: L %d taps %d\n", s[1], taps); ++bad; }
            }
        }
    }
    // padded and unpadded activation rows: the same step bit for bit (the padding only moves rows apart)
    for (const auto& s : shapes) {
        if (s[0] != 0) continue;
        std::vector<float> wa, wb;
        bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], false, &wa, 12345u);
        bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &wb, 12345u);
        if (wa.size() != wb.size() || std::memcmp(wa.data(), wb.data(), wa.size() * sizeof(float)) != 0) { std::printf("padded / plain rows differ: L %d F %d\n", s[1], s[3]); ++bad; }
        if (s[3] & (s[3] - 1)) continue;                   // rotated rows: power-of-two filter counts
        std::vector<float> wc;
        bad += train_case(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], true, &wc, 12345u, true);
        if (wa.size() != wc.size() || std::memcmp(wa.data(), wc.data(), wa.size() * sizeof(float)) != 0) { std::printf("rotated rows differ: L %d F %d\n", s[1], s[3]); ++bad; }
    }
    for (int trial = 0; trial < 215; ++trial) {
        const int nsym = (trial % 3 == 0) ? 2 : ((trial % 3 == 1) ? 4 : 20);
        std::vector<unsigned char> a(rnd() % (trial < 200 ? 200 : 1700)), b(rnd() % (trial < 200 ? 200 : 900));
        for (auto& c : a) c = (unsigned char)(65 + rnd() % nsym);
        for (auto& c : b) c = (unsigned char)(65 + rnd() % nsym);
        const int want = dp(a, b);
        if (strips<1>(a, b) != want || strips<12>(a, b) != want) { std::printf("strip mismatch at trial %d\n", trial); ++bad; }
        if (a.size() <= 768) {
            std::vector<uint64_t> peq(256 * 12, 0);
            for (size_t i = 0; i < a.size(); ++i) peq[a[i] * 12 + (i >> 6)] |= 1ull << (i & 63);
            const int got = fx_myers_distance<12>((int)a.size(), (int)b.size(), [&](int c, int w) { return peq[c * 12 + w]; },
                                                  [&](int i) { return (int)b[i]; });
            if (got != want) { std::printf("register form mismatch at trial %d\n", trial); ++bad; }
        }
    }
    // csrc/train_core.h fxt_lay / fxt_image_off: the LDS image of a member's weights (conv-kernel rows padded against bank conflicts)
    // is filled by the kernels from the Keras-order vector through fxt_image_off -- here every parameter's image offset is rebuilt
    // from the layout's own tables and compared, for filter counts that are powers of two (the shift) and that are not (the division)
    for (const auto& sh : std::vector<std::array<int, 6>>{{0, 8, 4, 32, 100, 5}, {0, 14, 4, 24, 50, 5}, {0, 30, 20, 40, 64, 3}, {0, 9, 4, 8, 16, 3},
                                                          {0, 12, 20, 5, 7, 4}, {0, 10, 4, 16, 33, 7}, {1, 14, 4, 0, 100, 0}, {2, 30, 20, 0, 100, 0}}) {
        const FxtNet n = fxt_net(sh[0], sh[1], sh[2], sh[3], sh[4], sh[5]);
        for (const bool padded : {false, true}) {
            const FxtLay y = fxt_lay(n, padded);
            std::vector<int> want((size_t)n.P, -1);
            if (n.kind == 0) {
                const int rows[3] = {n.K * n.A, n.K * n.F, n.K3 * n.F};
                for (int c = 0; c < 3; ++c) {
                    for (int r = 0; r < rows[c]; ++r)
                        for (int x = 0; x < n.F; ++x) want[(size_t)n.off_cw[c] + (size_t)r * n.F + x] = y.cw[c] + r * y.ldw + x;
                    for (int x = 0; x < n.F; ++x) want[(size_t)n.off_cb[c] + x] = y.cb[c] + x;
                }
            }
            for (int i = 0; i < n.nl; ++i) {
                for (int x = 0; x < n.dim[i] * n.dim[i + 1]; ++x) want[(size_t)n.off_w[i] + x] = y.w[i] + x;
                for (int x = 0; x < n.dim[i + 1]; ++x) want[(size_t)n.off_b[i] + x] = y.b[i] + x;
            }
            std::vector<char> used((size_t)y.total, 0);
            for (int g = 0; g < n.P; ++g) {
                const int at = fxt_image_off(n, y, g);
                if (want[(size_t)g] < 0 || at != want[(size_t)g] || at < 0 || at >= y.total || used[(size_t)at]) {
                    std::printf("image offset: kind %d F %d padded %d parameter %d -> %d, want %d\n", n.kind, n.F, (int)padded, g, at, want[(size_t)g]);
                    ++bad;
                    break;
                }
                used[(size_t)at] = 1;
            }
            if (!padded && y.total != n.P) { std::printf("unpadded image is not the Keras vector\n"); ++bad; }
            if (padded && n.kind == 0 && (n.F % 8) == 0 && y.ldw != n.F + 4) { std::printf("conv rows not padded\n"); ++bad; }
        }
    }
    // csrc/myers.h fx_bounded_distance (the band of fx_cache_density's kernel): the pattern in an exact-size buffer of m bytes, the
    // text row in an exact-size buffer of L bytes (NUL-padded when shorter), as the kernel sees them; near pairs and unrelated ones
    for (int trial = 0; trial < 600; ++trial) {
        const int nsym = (trial % 2) ? 4 : 20;
        std::vector<unsigned char> a(rnd() % 40), b;
        for (auto& c : a) c = (unsigned char)(65 + rnd() % nsym);
        if (trial % 5 == 0) { b.resize(rnd() % 40); for (auto& c : b) c = (unsigned char)(65 + rnd() % nsym); }
        else {
            b = a;
            for (int e_ = (int)(rnd() % 5); e_ > 0; --e_) {
                const unsigned op = rnd() % 3;
                if (op == 0 && !b.empty()) b[rnd() % b.size()] = (unsigned char)(65 + rnd() % nsym);
                else if (op == 1 && !b.empty()) b.erase(b.begin() + (long)(rnd() % b.size()));
                else b.insert(b.begin() + (long)(rnd() % (b.size() + 1)), (unsigned char)(65 + rnd() % nsym));
            }
        }
        const int want = dp(a, b);
        const int L = (int)std::max<size_t>(std::max(a.size(), b.size()), 1);
        std::vector<unsigned char> qa((size_t)L, 0), tb((size_t)L, 0);              // rows exactly L bytes wide
        std::copy(a.begin(), a.end(), qa.begin());
        std::copy(b.begin(), b.end(), tb.begin());
        const int m = (int)a.size();
        const int g1 = fx_bounded_distance<1>(false, m, L, qa.data(), tb.data());
        const int g2 = fx_bounded_distance<2>(false, m, L, qa.data(), tb.data());
        const int g3 = fx_bounded_distance<3>(false, m, L, qa.data(), tb.data());
        if (g1 != std::min(want, 2) || g2 != std::min(want, 3) || g3 != std::min(want, 4)) { std::printf("band mismatch at trial %d: dp %d, got %d %d %d\n", trial, want, g1, g2, g3); ++bad; }
        if (a.size() == b.size()) {
            int h = 0;
            for (size_t i = 0; i < a.size(); ++i) h += a[i] != b[i];
            if (fx_bounded_distance<2>(true, m, (int)a.size(), qa.data(), tb.data()) != std::min(h, 3)) { std::printf("hamming band mismatch at trial %d\n", trial); ++bad; }
        }
    }
    // fx_collect_lines on an answer array of exactly N words (64-byte aligned, as FxMailOut's rows are): whole lines only, never a
    // word past the end whatever has arrived
    for (int N : {1, 7, 8, 9, 16, 63, 64, 100, 4096}) {
        unsigned long long* ans = static_cast<unsigned long long*>(std::aligned_alloc(64, ((size_t)N * 8 + 63) / 64 * 64));
        std::vector<float> out((size_t)N, 7.f);
        const unsigned seq = 0x2345678u;
        for (int arrived : {0, N / 3, N - 1, N}) {
            for (int i = 0; i < N; ++i) ans[i] = ((unsigned long long)(i < arrived ? seq : seq - 1) << 32) | (unsigned)i;
            int flag = 0;
            const int64_t stop = fx_collect_lines(ans, out.data(), 0, N, seq, 64, &flag);
            const int64_t want_stop = (int64_t)(std::min(arrived, N) / 8) * 8;
            if (stop != want_stop && stop != 0) { std::printf("collect_lines: N %d arrived %d stopped at %lld\n", N, arrived, (long long)stop); ++bad; }
            for (int64_t i = 0; i < stop; ++i) { unsigned u; std::memcpy(&u, &out[(size_t)i], 4); if (u != (unsigned)i) { ++bad; break; } }
        }
        std::free(ans);
    }
    std::printf(bad ? "FAILED %d\n" : "sanitize_host: ok\n", bad);
    return bad != 0;
}
