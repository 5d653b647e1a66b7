// C ABI of libflexs_amd.so, part 5 of 6 (fx_internal.h): NoisyAbstractModel -- neighbour search, device-resident cache,
// distances / densities, table landscapes, the blend.
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

extern "C" {

// ----------------------------------------------------- NoisyAbstractModel
static int min_dist_common(fx_engine* e, int mode, const uint8_t* queries, int64_t Q, const uint8_t* d_cache,
                           int64_t C, int L, int32_t* dist, int64_t* argmin) {
    if (mode != FX_LEVENSHTEIN && mode != FX_HAMMING) return fx_fail(e, FX_EINVAL, "unknown distance mode");
    if (Q == 0) return FX_OK;
    if (C == 0) {                                           // noisy_abstract_model.py:44-45
        for (int64_t i = 0; i < Q; ++i) { dist[i] = 0; argmin[i] = -1; }
        return FX_OK;
    }
    e->counters.pair_evals += Q * C;
    int rc;
    if ((size_t)Q * L + (size_t)Q * 12 <= (size_t)e->zero_copy_bytes) {
        // explorer-size query batch: queries read from / results written to mapped pinned memory (the keys stay on the device)
        FxZeroCopy z;
        void* d_keys = nullptr;
        if ((rc = fx_zero_copy_buffers(e, (size_t)Q * L + 16, (size_t)Q * 16, &z))) return rc;
        if ((rc = fx_scratch(e, 1, (size_t)Q * 8, &d_keys))) return rc;
        std::memcpy(z.h_in, queries, (size_t)Q * L);
        int64_t* zd_arg = (int64_t*)z.d_out;
        int32_t* zd_dist = (int32_t*)(z.d_out + (size_t)Q * 8);
        if ((rc = fx_launch_min_dist(e, mode, (const uint8_t*)z.d_in, Q, d_cache, C, L, (unsigned long long*)d_keys))) return rc;
        if ((rc = fx_launch_min_dist_finish(e, (unsigned long long*)d_keys, Q, C, zd_dist, zd_arg))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(argmin, z.h_out, (size_t)Q * 8);
        std::memcpy(dist, z.h_out + (size_t)Q * 8, (size_t)Q * 4);
        return FX_OK;
    }
    for (int64_t q0 = 0; q0 < Q; q0 += 32768) {
        const int64_t qn = std::min<int64_t>(32768, Q - q0);
        void *d_q = nullptr, *d_res = nullptr;
        if ((rc = fx_scratch(e, 0, (size_t)qn * L + 16, &d_q))) return rc;
        if ((rc = fx_scratch(e, 1, (size_t)qn * 24, &d_res))) return rc;
        unsigned long long* d_keys = (unsigned long long*)d_res;
        int64_t* d_arg = (int64_t*)((char*)d_res + (size_t)qn * 8);
        int32_t* d_dist = (int32_t*)((char*)d_res + (size_t)qn * 16);
        FX_HIP(e, hipMemcpyAsync(d_q, queries + q0 * L, (size_t)qn * L, hipMemcpyHostToDevice, e->stream));
        if ((rc = fx_launch_min_dist(e, mode, (const uint8_t*)d_q, qn, d_cache, C, L, d_keys))) return rc;
        if ((rc = fx_launch_min_dist_finish(e, d_keys, qn, C, d_dist, d_arg))) return rc;
        FX_HIP(e, hipMemcpyAsync(dist + q0, d_dist, (size_t)qn * 4, hipMemcpyDeviceToHost, e->stream));
        FX_HIP(e, hipMemcpyAsync(argmin + q0, d_arg, (size_t)qn * 8, hipMemcpyDeviceToHost, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    return FX_OK;
}

int fx_min_dist(fx_engine* e, int mode, const uint8_t* queries, int64_t Q, const uint8_t* cache, int64_t C, int L,
                int32_t* dist, int64_t* argmin) {
    if (!e || Q < 0 || C < 0 || L < 0) return FX_EINVAL;
    if (Q > 0 && (!queries || !dist || !argmin)) return fx_fail(e, FX_EINVAL, "null buffer");
    if (C > 0 && !cache) return fx_fail(e, FX_EINVAL, "null cache buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    void* d_cache = nullptr;
    if (C > 0 && Q > 0) {
        int rc = fx_scratch(e, 2, (size_t)C * L + 16, &d_cache);
        if (rc) return rc;
        FX_HIP(e, hipMemcpyAsync(d_cache, cache, (size_t)C * L, hipMemcpyHostToDevice, e->stream));
    }
    return min_dist_common(e, mode, queries, Q, (const uint8_t*)d_cache, C, L, dist, argmin);
}

int fx_cache_create(fx_engine* e, int L, fx_cache** out) {
    if (!e || !out || L < 0) return FX_EINVAL;
    fx_cache* c = new (std::nothrow) fx_cache();
    if (!c) return FX_ENOMEM;
    c->eng = e; c->L = L;
    *out = c;
    return FX_OK;
}

int fx_cache_destroy(fx_cache* c) {
    if (!c) return FX_OK;
    (void)hipSetDevice(c->eng->device);
    (void)hipStreamSynchronize(c->eng->stream);
    if (c->d_keys) (void)hipFree(c->d_keys);
    delete c;
    return FX_OK;
}

int64_t fx_cache_size(const fx_cache* c) { return c ? c->size : FX_EINVAL; }

int fx_cache_append(fx_cache* c, const uint8_t* keys, int64_t n) {
    if (!c || n < 0) return FX_EINVAL;
    if (n == 0) return FX_OK;
    fx_engine* e = c->eng;
    if (!keys) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const size_t rowb = (size_t)std::max(c->L, 1);
    if (c->size + n > c->capacity) {
        int64_t cap = std::max<int64_t>(c->capacity * 2, std::max<int64_t>(c->size + n, 4096));
        uint8_t* nk = nullptr;
        if (hipMalloc(&nk, (size_t)cap * rowb + 16) != hipSuccess) { (void)hipGetLastError(); return fx_fail(e, FX_ENOMEM, "cache grow failed"); }
        FX_HIP(e, hipStreamSynchronize(e->stream));
        if (c->size) FX_HIP(e, hipMemcpy(nk, c->d_keys, (size_t)c->size * rowb, hipMemcpyDeviceToDevice));
        if (c->d_keys) FX_HIP(e, hipFree(c->d_keys));
        c->d_keys = nk; c->capacity = cap;
    }
    FX_HIP(e, hipMemcpy(c->d_keys + (size_t)c->size * rowb, keys, (size_t)n * rowb, hipMemcpyHostToDevice));
    c->size += n;
    return FX_OK;
}

int fx_cache_min_dist(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, int32_t* dist, int64_t* argmin) {
    if (!c || Q < 0) return FX_EINVAL;
    fx_engine* e = c->eng;
    if (Q > 0 && (!queries || !dist || !argmin)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    return min_dist_common(e, mode, queries, Q, c->d_keys, c->size, c->L, dist, argmin);
}

int fx_cache_nam_query(fx_cache* c, fx_table* t, int bits, const uint8_t lut[256], int mode, const uint8_t* append_keys,
                       int64_t n_append, const uint8_t* queries, int64_t Q, const double* E, const double* alpha_tab, int n_tab,
                       double* out, int32_t* dist, int64_t* argmin, int32_t* flags) {
    if (!c || !t || Q < 0 || n_append < 0 || n_tab < 1 || !lut || bits < 1 || bits > 8) return FX_EINVAL;
    if (n_append > 0 && !append_keys) return FX_EINVAL;
    fx_engine* e = c->eng;
    if (t->eng != e) return fx_fail(e, FX_EINVAL, "table and cache belong to different engines");
    if ((int64_t)bits * c->L > 40) return fx_fail(e, FX_EINVAL, "sequence too long for a packed-k-mer table");
    if (mode != FX_LEVENSHTEIN && mode != FX_HAMMING) return fx_fail(e, FX_EINVAL, "unknown distance mode");
    if (Q == 0) return n_append ? fx_cache_append(c, append_keys, n_append) : FX_OK;
    if (Q > 32768) return fx_fail(e, FX_EUNSUPPORTED, "fx_cache_nam_query: at most 32768 queries per call (fx_cache_min_dist batches larger sets)");
    if (!queries || !E || !alpha_tab || !out || !dist || !argmin || !flags) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const int L = c->L;
    // the keys cached since the last call (the previous batch's sequences) join the cache first: in the same submission when
    // they fit the staging area and the cache has room, else by fx_cache_append (a synchronous copy)
    const size_t app_bytes = (size_t)n_append * std::max(L, 1);
    const bool app_inline = n_append > 0 && c->size + n_append <= c->capacity &&
                            app_bytes + (size_t)Q * (L + 36) + (size_t)n_tab * 8 + 64 <= (size_t)e->zero_copy_bytes;
    if (n_append > 0 && !app_inline) {
        const int rc0 = fx_cache_append(c, append_keys, n_append);
        if (rc0) return rc0;
    }
    // inputs: queries | E | alpha table; outputs: out | argmin | dist | flags -- mapped pinned memory when the batch is of
    // explorer size, device scratch + copies otherwise; the neighbour keys never leave the device
    const size_t o_E = ((size_t)Q * L + 15) / 16 * 16, o_tab = o_E + (size_t)Q * 8, o_app = o_tab + ((size_t)n_tab * 8 + 15) / 16 * 16,
                 in_bytes = o_app + (app_inline ? app_bytes : 0);
    const size_t o_arg = (size_t)Q * 8, o_dist = 2 * (size_t)Q * 8, o_flags = o_dist + (size_t)Q * 4, out_bytes = o_flags + (size_t)Q * 4;
    int rc;
    char *h_in = nullptr, *d_in = nullptr, *h_out = nullptr, *d_out = nullptr;
    const bool zc = in_bytes + out_bytes <= (size_t)e->zero_copy_bytes;
    FxZeroCopy z{};
    if ((rc = fx_zero_copy_buffers(e, in_bytes, out_bytes, &z))) return rc;     // (staging for the copy path too)
    h_in = z.h_in; h_out = z.h_out;
    if (zc) { d_in = z.d_in; d_out = z.d_out; }
    else {
        void *p0 = nullptr, *p1 = nullptr;
        if ((rc = fx_scratch(e, 0, in_bytes + 16, &p0))) return rc;
        if ((rc = fx_scratch(e, 3, out_bytes + 16, &p1))) return rc;
        d_in = (char*)p0; d_out = (char*)p1;
    }
    std::memcpy(h_in, queries, (size_t)Q * L);
    std::memcpy(h_in + o_E, E, (size_t)Q * 8);
    std::memcpy(h_in + o_tab, alpha_tab, (size_t)n_tab * 8);
    if (app_inline) {
        std::memcpy(h_in + o_app, append_keys, app_bytes);
        FX_HIP(e, hipMemcpyAsync(c->d_keys + (size_t)c->size * std::max(L, 1), h_in + o_app, app_bytes, hipMemcpyHostToDevice, e->stream));
        c->size += n_append;
    }
    const int64_t C_ = c->size;
    if (!zc) FX_HIP(e, hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    int64_t* d_arg = (int64_t*)(d_out + o_arg);
    int32_t* d_dist = (int32_t*)(d_out + o_dist);
    if (C_ > 0) {
        void* d_keys = nullptr;
        if ((rc = fx_scratch(e, 1, (size_t)Q * 8, &d_keys))) return rc;
        e->counters.pair_evals += Q * C_;
        if ((rc = fx_launch_min_dist(e, mode, (const uint8_t*)d_in, Q, c->d_keys, C_, L, (unsigned long long*)d_keys))) return rc;
        if ((rc = fx_launch_min_dist_finish(e, (unsigned long long*)d_keys, Q, C_, d_dist, d_arg))) return rc;
    } else {
        FX_HIP(e, hipMemsetAsync(d_dist, 0, (size_t)Q * 4, e->stream));            // noisy_abstract_model.py:44-45
        FX_HIP(e, hipMemsetAsync(d_arg, 0xFF, (size_t)Q * 8, e->stream));          // -1: the query itself
    }
    if ((rc = fx_launch_nam_table_blend(e, Q, (const uint8_t*)d_in, c->d_keys, d_arg, d_dist, t->d_table, t->len, L, bits,
                                        (const double*)(d_in + o_E), (const double*)(d_in + o_tab), n_tab, (double*)d_out,
                                        (int32_t*)(d_out + o_flags)))) return rc;
    if (!zc) FX_HIP(e, hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
    std::memcpy(out, h_out, (size_t)Q * 8);
    std::memcpy(argmin, h_out + o_arg, (size_t)Q * 8);
    std::memcpy(dist, h_out + o_dist, (size_t)Q * 4);
    std::memcpy(flags, h_out + o_flags, (size_t)Q * 4);
    return FX_OK;
}

static int cache_distances(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, uint8_t* out, int bound);
int fx_cache_distances(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, uint8_t* out) {
    return cache_distances(c, mode, queries, Q, out, 0);
}
// bound > 0: out = min(distance, bound + 1) where the banded kernel applies (explorer-size calls), the exact matrix otherwise
static int cache_distances(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, uint8_t* out, int bound) {
    if (!c || Q < 0) return FX_EINVAL;
    fx_engine* e = c->eng;
    if (mode != FX_LEVENSHTEIN && mode != FX_HAMMING) return fx_fail(e, FX_EINVAL, "unknown distance mode");
    if (Q == 0 || c->size == 0) return FX_OK;
    if (!queries || !out) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    int rc;
    if ((size_t)Q * c->L + (size_t)Q * c->size <= (size_t)e->zero_copy_bytes) {
        // a DyNA-PPO environment step (one sequence against everything seen): query and distance row in mapped pinned memory
        FxZeroCopy z;
        if ((rc = fx_zero_copy_buffers(e, (size_t)Q * c->L + 16, (size_t)Q * c->size, &z))) return rc;
        std::memcpy(z.h_in, queries, (size_t)Q * c->L);
        rc = bound > 0 ? fx_launch_distances_bounded(e, mode, (const uint8_t*)z.d_in, Q, c->d_keys, c->size, c->L, bound, (uint8_t*)z.d_out) : FX_EUNSUPPORTED;
        if (rc == FX_EUNSUPPORTED) rc = fx_launch_distances(e, mode, (const uint8_t*)z.d_in, Q, c->d_keys, c->size, c->L, (uint8_t*)z.d_out);
        if (rc) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(out, z.h_out, (size_t)Q * c->size);
        return FX_OK;
    }
    const int64_t qstep = std::max<int64_t>(1, std::min<int64_t>(32768, ((int64_t)1 << 28) / std::max<int64_t>(c->size, 1)));
    for (int64_t q0 = 0; q0 < Q; q0 += qstep) {
        const int64_t qn = std::min<int64_t>(qstep, Q - q0);
        void *d_q = nullptr, *d_out = nullptr;
        if ((rc = fx_scratch(e, 0, (size_t)qn * c->L + 16, &d_q))) return rc;
        if ((rc = fx_scratch(e, 1, (size_t)qn * c->size, &d_out))) return rc;
        FX_HIP(e, hipMemcpyAsync(d_q, queries + q0 * c->L, (size_t)qn * c->L, hipMemcpyHostToDevice, e->stream));
        if ((rc = fx_launch_distances(e, mode, (const uint8_t*)d_q, qn, c->d_keys, c->size, c->L, (uint8_t*)d_out))) return rc;
        FX_HIP(e, hipMemcpyAsync(out + q0 * c->size, d_out, (size_t)qn * c->size, hipMemcpyDeviceToHost, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    return FX_OK;
}

int fx_cache_density(fx_cache* c, int mode, const uint8_t* queries, int64_t Q, int radius, const double* fitness, double* density,
                     int32_t* neighbours) {
    if (!c || Q < 0 || radius < 0) return FX_EINVAL;
    fx_engine* e = c->eng;
    if (mode != FX_LEVENSHTEIN && mode != FX_HAMMING) return fx_fail(e, FX_EINVAL, "unknown distance mode");
    if (Q == 0) return FX_OK;
    if (!queries || !density || !neighbours || (!fitness && c->size > 0)) return fx_fail(e, FX_EINVAL, "null buffer");
    const int64_t C = c->size;
    for (int64_t q = 0; q < Q; ++q) { density[q] = 0.0; neighbours[q] = 0; }
    if (C == 0) return FX_OK;
    // query blocks whose distance rows fit the pinned staging area; a row at a time on the host: a byte compare per key, the
    // division and the sum only for the few neighbours -- same operations, same order as `for s in all_seqs: ... dens += f / dist`
    const int64_t qstep = std::max<int64_t>(1, std::min<int64_t>(Q, (int64_t)(e->zero_copy_bytes > 0 ? e->zero_copy_bytes : (1 << 18)) / std::max<int64_t>(C + c->L, 1)));
    std::vector<uint8_t> rows;
    const int r = radius > 254 ? 254 : radius;
    for (int64_t q0 = 0; q0 < Q; q0 += qstep) {
        const int64_t qn = std::min<int64_t>(qstep, Q - q0);
        rows.resize((size_t)qn * (size_t)C);
        if (int rc = cache_distances(c, mode, queries + q0 * c->L, qn, rows.data(), r)) return rc;
        for (int64_t q = 0; q < qn; ++q) {
            const uint8_t* d = rows.data() + (size_t)q * (size_t)C;
            double dens = 0.0;
            int32_t cnt = 0;
            for (int64_t i = 0; i < C; ++i) {
                const unsigned di = d[i];
                if ((unsigned)(di - 1u) < (unsigned)r) { dens += fitness[i] / (double)di; ++cnt; }    // 0 < dist <= radius
            }
            density[q0 + q] = dens;
            neighbours[q0 + q] = cnt;
        }
    }
    return FX_OK;
}

int fx_table_create(fx_engine* e, const double* table, int64_t len, fx_table** out) {
    if (!e || !table || len < 1 || !out) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    fx_table* t = new (std::nothrow) fx_table();
    if (!t) return FX_ENOMEM;
    t->eng = e; t->len = len;
    if (hipMalloc(&t->d_table, sizeof(double) * (size_t)len) != hipSuccess) { (void)hipGetLastError(); delete t; return fx_fail(e, FX_ENOMEM, "table alloc failed"); }
    FX_HIP(e, hipMemcpy(t->d_table, table, sizeof(double) * (size_t)len, hipMemcpyHostToDevice));
    *out = t;
    return FX_OK;
}

int fx_table_destroy(fx_table* t) {
    if (!t) return FX_OK;
    (void)hipSetDevice(t->eng->device);
    (void)hipStreamSynchronize(t->eng->stream);
    if (t->d_table) (void)hipFree(t->d_table);
    delete t;
    return FX_OK;
}

int fx_table_lookup(fx_table* t, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], int bits, double* out) {
    if (!t || N < 0 || L < 0 || !lut || bits < 1 || bits > 8 || (int64_t)bits * L > 40) return FX_EINVAL;
    fx_engine* e = t->eng;
    if (N == 0) return FX_OK;
    if (!ascii || !out) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((size_t)N * L + (size_t)N * 8 <= (size_t)e->zero_copy_bytes) {
        FxZeroCopy z;
        if ((rc = fx_zero_copy_buffers(e, (size_t)N * L + 16, (size_t)N * 8, &z))) return rc;
        std::memcpy(z.h_in, ascii, (size_t)N * L);
        if ((rc = fx_upload_lut(e, lut))) return rc;
        if ((rc = fx_launch_table_lookup(e, t->d_table, t->len, (const uint8_t*)z.d_in, N, L, bits, (double*)z.d_out))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(out, z.h_out, (size_t)N * 8);
        return FX_OK;
    }
    if ((rc = fx_scratch(e, 0, (size_t)N * L + 16, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, (size_t)N * 8, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, ascii, (size_t)N * L, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_table_lookup(e, t->d_table, t->len, (const uint8_t*)d_in, N, L, bits, (double*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(out, d_out, (size_t)N * 8, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

int fx_table_additive(fx_table* t, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], int ncol, double* out) {
    if (!t || N < 0 || L < 1 || !lut || ncol < 1 || ncol > 256) return FX_EINVAL;
    fx_engine* e = t->eng;
    if ((int64_t)L * ncol != t->len) return fx_fail(e, FX_ESHAPE, "additive table must hold L x ncol entries");
    for (int c = 0; c < 256; ++c)
        if (lut[c] >= ncol) return fx_fail(e, FX_EINVAL, "additive table: lut entry outside [0, ncol)");
    if (N == 0) return FX_OK;
    if (!ascii || !out) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((size_t)N * L + (size_t)N * 8 <= (size_t)e->zero_copy_bytes) {
        FxZeroCopy z;
        if ((rc = fx_zero_copy_buffers(e, (size_t)N * L + 16, (size_t)N * 8, &z))) return rc;
        std::memcpy(z.h_in, ascii, (size_t)N * L);
        if ((rc = fx_upload_lut(e, lut))) return rc;
        if ((rc = fx_launch_additive_sum(e, t->d_table, L, ncol, (const uint8_t*)z.d_in, N, (double*)z.d_out))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(out, z.h_out, (size_t)N * 8);
        return FX_OK;
    }
    if ((rc = fx_scratch(e, 0, (size_t)N * L + 16, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, (size_t)N * 8, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, ascii, (size_t)N * L, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_additive_sum(e, t->d_table, L, ncol, (const uint8_t*)d_in, N, (double*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(out, d_out, (size_t)N * 8, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

int fx_nam_combine(fx_engine* e, int64_t Q, const double* signal, const double* noise, const int32_t* d,
                   const double* alpha_tab, int n_tab, double* out) {
    if (!e || Q < 0 || n_tab < 1) return FX_EINVAL;
    if (Q == 0) return FX_OK;
    if (!signal || !noise || !d || !alpha_tab || !out) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const size_t qb = (size_t)Q * 8;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if (3 * qb + (size_t)Q * 4 + (size_t)n_tab * 8 <= (size_t)e->zero_copy_bytes) {
        FxZeroCopy z;
        if ((rc = fx_zero_copy_buffers(e, 2 * qb + (size_t)n_tab * 8 + (size_t)Q * 4, qb, &z))) return rc;
        const size_t o_tab = 2 * qb, o_d = 2 * qb + (size_t)n_tab * 8;
        std::memcpy(z.h_in, signal, qb);
        std::memcpy(z.h_in + qb, noise, qb);
        std::memcpy(z.h_in + o_tab, alpha_tab, (size_t)n_tab * 8);
        std::memcpy(z.h_in + o_d, d, (size_t)Q * 4);
        if ((rc = fx_launch_nam_combine(e, Q, (const double*)z.d_in, (const double*)(z.d_in + qb), (const int32_t*)(z.d_in + o_d),
                                        (const double*)(z.d_in + o_tab), n_tab, (double*)z.d_out))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(out, z.h_out, qb);
        return FX_OK;
    }
    if ((rc = fx_scratch(e, 0, 2 * qb + (size_t)Q * 4 + (size_t)n_tab * 8 + 64, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, qb, &d_out))) return rc;
    char* base = (char*)d_in;
    double* d_sig = (double*)base;
    double* d_noi = (double*)(base + qb);
    double* d_tab = (double*)(base + 2 * qb);
    int32_t* d_d = (int32_t*)(base + 2 * qb + (size_t)n_tab * 8);
    FX_HIP(e, hipMemcpyAsync(d_sig, signal, qb, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_noi, noise, qb, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_tab, alpha_tab, (size_t)n_tab * 8, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_d, d, (size_t)Q * 4, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_launch_nam_combine(e, Q, d_sig, d_noi, d_d, d_tab, n_tab, (double*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(out, d_out, qb, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

}  // extern "C"
