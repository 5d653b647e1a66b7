// C ABI of libflexs_amd.so, part 6 of 6 (fx_internal.h): test and profiling hooks (fx_debug_*).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

#include "fx_common.h"
#include "fx_internal.h"
#include "myers.h"
#include "np_sum.h"
#include <atomic>
#include <mutex>
#include <chrono>


// strips of `lw_rows` = 64 x LW pattern rows, as k_min_dist_long runs them for patterns beyond 768 symbols (LW = 12 there;
// the test hook also takes LW = 1 so that short strings cross many strip boundaries)
template <int LW>
static int myers_strips_host(const uint8_t* a, int la, const uint8_t* b, int lb) {
    std::vector<signed char> h((size_t)std::max(lb, 1), 0);
    const int nstrips = la > 0 ? (la + 64 * LW - 1) / (64 * LW) : 1;
    int part = 0;
    for (int s = 0; s < nstrips; ++s) {
        const int r0 = s * 64 * LW, rows = std::min(la - r0, 64 * LW);
        std::vector<uint64_t> peq((size_t)256 * LW, 0);
        for (int i = 0; i < rows; ++i) peq[(size_t)a[r0 + i] * LW + (i >> 6)] |= 1ull << (i & 63);
        part = fx_myers_strip<LW>(rows > 0 ? rows : 0, lb, [&](int c, int w) { return peq[(size_t)c * LW + w]; },
                                  [&](int i) { return (int)b[i]; }, h.data(), h.data(), 1, s == 0, s == nstrips - 1);
    }
    return la + part;
}


extern "C" {

// --------------------------------------------------------------- debug / test
// Host-only helpers (no device needed) exported so the CPU test-suite can check
// the weight packing and the bit-parallel distance without a GPU.
int64_t fx_debug_packed_size(int kind, int L, int A, int F, int H, int K) {
    return fx_pack_layout(FxShape{kind, L, A, F, H, K}).alloc_floats;
}
int fx_debug_trace_read(fx_engine* e, uint64_t* out, int64_t cap_words) {
    if (!e || !out || cap_words < 0) return FX_EINVAL;
    if (!e->d_trace) return fx_fail(e, FX_ESTATE, "no trace was recorded (set the \"trace\" option before scoring)");
    FX_HIP(e, hipSetDevice(e->device));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    const size_t n = std::min<size_t>((size_t)cap_words * 8, FX_TRACE_BYTES);
    FX_HIP(e, hipMemcpy(out, e->d_trace, n, hipMemcpyDeviceToHost));
    return FX_OK;
}
int fx_debug_time_score(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                        const uint8_t lut[256], float* d_planes, int64_t stride, int reps, float* total_ms) {
    if (!e || !total_ms || reps < 1) return FX_EINVAL;
    int rc = fx_score_planes_dev(e, models, M, d_ascii, N, L, lut, d_planes, stride);   // validates; warms the caches
    if (rc) return rc;
    FX_HIP(e, hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; ++i)
        if ((rc = fx_score_planes_dev(e, models, M, d_ascii, N, L, lut, d_planes, stride))) return rc;
    FX_HIP(e, hipEventRecord(e->ev1, e->stream));
    FX_HIP(e, hipEventSynchronize(e->ev1));
    FX_HIP(e, hipEventElapsedTime(total_ms, e->ev0, e->ev1));
    return FX_OK;
}
int fx_debug_train_trace(fx_engine* e, uint64_t* out64) {
    if (!e || !out64) return FX_EINVAL;
    if (!e->d_train_dbg) return fx_fail(e, FX_ESTATE, "no training trace: set the option train_trace and run fx_train_fit");
    FX_HIP(e, hipSetDevice(e->device));
    FX_HIP(e, hipMemcpy(out64, e->d_train_dbg, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FX_OK;
}
int fx_debug_time_min_dist(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, int reps, float* total_ms) {
    if (!c || !queries || !total_ms || reps < 1 || Q < 1 || Q > 32768) return FX_EINVAL;
    fx_engine* e = c->eng;
    if (mode != FX_LEVENSHTEIN && mode != FX_HAMMING) return fx_fail(e, FX_EINVAL, "unknown distance mode");
    if (c->size == 0) return fx_fail(e, FX_EINVAL, "empty cache");
    FX_HIP(e, hipSetDevice(e->device));
    void *d_q = nullptr, *d_res = nullptr;
    int rc;
    if ((rc = fx_scratch(e, 0, (size_t)Q * c->L + 16, &d_q))) return rc;
    if ((rc = fx_scratch(e, 1, (size_t)Q * 24, &d_res))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_q, queries, (size_t)Q * c->L, hipMemcpyHostToDevice, e->stream));
    unsigned long long* d_keys = (unsigned long long*)d_res;
    if ((rc = fx_launch_min_dist(e, mode, (const uint8_t*)d_q, Q, c->d_keys, c->size, c->L, d_keys))) return rc;   // warm
    FX_HIP(e, hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; ++i)          // key reset + K4, what one neighbour search enqueues (the finish kernel is O(Q))
        if ((rc = fx_launch_min_dist(e, mode, (const uint8_t*)d_q, Q, c->d_keys, c->size, c->L, d_keys))) return rc;
    FX_HIP(e, hipEventRecord(e->ev1, e->stream));
    FX_HIP(e, hipEventSynchronize(e->ev1));
    FX_HIP(e, hipEventElapsedTime(total_ms, e->ev0, e->ev1));
    return FX_OK;
}
int64_t fx_debug_mfma_per_tile(int kind, int L, int A, int F, int H, int K) {
    if (kind < FX_CNN || kind > FX_GE || L < 1 || A < 2 || H < 1) return FX_EINVAL;
    return fx_mfma_per_tile(FxShape{kind, L, A, kind == FX_CNN ? F : 0, H, kind == FX_CNN ? K : 0});
}
int fx_debug_pack_layout(int kind, int L, int A, int F, int H, int K, int64_t* out16) {
    if (!out16) return FX_EINVAL;
    const FxPackLayout p = fx_pack_layout(FxShape{kind, L, A, F, H, K});
    const int64_t v[16] = {p.FT, p.HT, p.SG1, p.off_first, p.off_c2, p.off_c3, p.off_cb, p.conv_floats,
                           p.off_d1, p.off_d2, p.off_d3, p.off_db, p.RLH, p.total_floats, p.off_w1p, p.HTR};
    std::memcpy(out16, v, sizeof(v));
    return FX_OK;
}
int fx_debug_pack_weights(int kind, int L, int A, int F, int H, int K, const float* blob, int64_t n, float* packed,
                          int64_t cap) {
    const FxShape s{kind, L, A, F, H, K};
    if (!blob || !packed || n != fx_num_params(s) || cap < fx_pack_layout(s).alloc_floats) return FX_EINVAL;
    fx_pack_weights(s, blob, packed);
    return FX_OK;
}
int fx_debug_mfma_probe(fx_engine* e, const float* a64, const float* b64, const float* c256, float* d256) {
    if (!e || !a64 || !b64 || !c256 || !d256) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    void* buf = nullptr;
    int rc = fx_scratch(e, 0, sizeof(float) * (64 + 64 + 256 + 256), &buf);
    if (rc) return rc;
    float* d = (float*)buf;
    FX_HIP(e, hipMemcpyAsync(d, a64, 256, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d + 64, b64, 256, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d + 128, c256, 1024, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_launch_mfma_probe(e, d, d + 64, d + 128, d + 384))) return rc;
    FX_HIP(e, hipMemcpyAsync(d256, d + 384, 1024, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}
int fx_debug_myers_strips(const uint8_t* a, int la, const uint8_t* b, int lb, int words_per_strip) {
    if (la < 0 || lb < 0 || (la > 0 && !a) || (lb > 0 && !b)) return -1;
    for (int i = 0; i < lb; ++i) if (b[i] == 0) return -1;       // (the strip routine stops at a NUL: not part of any alphabet)
    if (words_per_strip == 1) return myers_strips_host<1>(a, la, b, lb);
    if (words_per_strip == 12) return myers_strips_host<12>(a, la, b, lb);
    return -1;
}
int fx_debug_bounded_distance(const uint8_t* a, int la, const uint8_t* b, int lb, int K, int hamming) {
    // the band of k_distances_bounded (myers.h) on the host: pattern a against the text row b, both NUL-padded to a common width
    if (la < 0 || lb < 0 || K < 1 || K > 3 || (la > 0 && !a) || (lb > 0 && !b)) return -1;
    const int L = std::max(std::max(la, lb), 1);
    std::vector<uint8_t> qa((size_t)L + 1, 0), tb((size_t)L + 1, 0);
    if (la) std::memcpy(qa.data(), a, (size_t)la);
    if (lb) std::memcpy(tb.data(), b, (size_t)lb);
    if (K == 1) return fx_bounded_distance<1>(hamming != 0, la, L, qa.data(), tb.data());
    if (K == 2) return fx_bounded_distance<2>(hamming != 0, la, L, qa.data(), tb.data());
    return fx_bounded_distance<3>(hamming != 0, la, L, qa.data(), tb.data());
}
int fx_debug_myers(const uint8_t* a, int la, const uint8_t* b, int lb) {
    // pattern = a, text = b; same code path as the device kernel (myers.h)
    if (la < 0 || lb < 0) return -1;
    if (la > 768) return myers_strips_host<12>(a, la, b, lb);   // what k_min_dist_long runs
    if (la <= 32) {                                   // the kernels' 32-bit single-word form (mindist.hip)
        uint32_t peq32[256] = {0};
        for (int i = 0; i < la; ++i) peq32[a[i]] |= 1u << i;
        return fx_myers_distance<1, false, uint32_t>(
            la, lb, [&](int c, int) { return peq32[c]; }, [&](int i) { return (int)b[i]; });
    }
    static thread_local uint64_t peq[256 * 12];
    std::memset(peq, 0, sizeof(peq));
    for (int i = 0; i < la; ++i) peq[a[i] * 12 + (i >> 6)] |= 1ull << (i & 63);
    auto pf = [&](int c, int w) { return peq[c * 12 + w]; };
    auto tf = [&](int i) { return (int)b[i]; };
    const int W = (la + 63) / 64;
    switch (W) {                                      // the same instantiations the kernels use
        case 0:
        case 1: return fx_myers_distance<1>(la, lb, pf, tf);
        case 2: return fx_myers_distance<2>(la, lb, pf, tf);
        case 3: return fx_myers_distance<3>(la, lb, pf, tf);
        case 4: return fx_myers_distance<4>(la, lb, pf, tf);
        case 5: case 6: return fx_myers_distance<6>(la, lb, pf, tf);
        case 7: case 8: return fx_myers_distance<8>(la, lb, pf, tf);
        default: return fx_myers_distance<12>(la, lb, pf, tf);
    }
}

}  // extern "C"
