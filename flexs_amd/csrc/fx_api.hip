// C ABI of libflexs_amd.so (see include/flexs_amd.h for the contract and the
// reference file:line each entry point replaces).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

#include "fx_common.h"
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


// ------------------------------------------------------------------ helpers
static thread_local std::string g_last_error_noengine;

int fx_fail(fx_engine* e, int status, const std::string& msg) {
    if (e) e->last_error = msg; else g_last_error_noengine = msg;
    return status;
}

int fx_scratch(fx_engine* e, int slot, size_t bytes, void** out) {
    if (bytes > e->scratch_bytes[slot]) {
        if (e->d_scratch[slot]) {
            FX_HIP(e, hipStreamSynchronize(e->stream));
            FX_HIP(e, hipFree(e->d_scratch[slot]));
            e->d_scratch[slot] = nullptr; e->scratch_bytes[slot] = 0;
        }
        size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        if (hipMalloc(&e->d_scratch[slot], cap) != hipSuccess) {
            (void)hipGetLastError();
            return fx_fail(e, FX_ENOMEM, "hipMalloc of device scratch failed");
        }
        e->scratch_bytes[slot] = cap;
    }
    *out = e->d_scratch[slot];
    return FX_OK;
}

int fx_zero_pool(fx_engine* e, size_t bytes, void** out) {
    if (bytes > e->zero_pool_bytes) {
        if (e->d_zero_pool) {
            FX_HIP(e, hipStreamSynchronize(e->stream));
            FX_HIP(e, hipFree(e->d_zero_pool));
            e->d_zero_pool = nullptr; e->zero_pool_bytes = 0;
        }
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        if (hipMalloc(&e->d_zero_pool, cap) != hipSuccess) {
            (void)hipGetLastError();
            return fx_fail(e, FX_ENOMEM, "hipMalloc of the zero pool failed");
        }
        e->zero_pool_bytes = cap;
        // ordered explicitly: the memset rides the stream the segment launches use AND is waited for, so the zeros are
        // there whatever stream (e->stream may be a lent torch stream) the next launch runs on
        FX_HIP(e, hipMemsetAsync(e->d_zero_pool, 0, cap, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    *out = e->d_zero_pool;
    return FX_OK;
}

int fx_pinned(fx_engine* e, int slot, size_t bytes, void** out) {
    if (bytes > e->pinned_bytes[slot]) {
        if (e->h_pinned[slot]) {
            FX_HIP(e, hipStreamSynchronize(e->stream));
            FX_HIP(e, hipHostFree(e->h_pinned[slot]));
            e->h_pinned[slot] = nullptr; e->pinned_bytes[slot] = 0;
        }
        size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        if (hipHostMalloc(&e->h_pinned[slot], cap, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            return fx_fail(e, FX_ENOMEM, "hipHostMalloc of pinned staging failed");
        }
        e->pinned_bytes[slot] = cap;
    }
    *out = e->h_pinned[slot];
    return FX_OK;
}

// Explorer-size calls of the small kernels (distances, blend, table look-ups): inputs and outputs in the engine's mapped
// pinned staging areas, read / written by the kernel directly -- one launch and one wait instead of 3-5 copy enqueues
// around them (a hipMemcpyAsync of a few hundred bytes costs as much as the kernel).  The kernel's end makes its
// stores to host memory visible; the host copies the result out after the stream wait.
struct FxZeroCopy { char *h_in, *d_in, *h_out, *d_out; };
static int fx_zero_copy_buffers(fx_engine* e, size_t in_bytes, size_t out_bytes, FxZeroCopy* z) {
    void *hi = nullptr, *ho = nullptr, *di = nullptr, *dout = nullptr;
    int rc;
    if ((rc = fx_pinned(e, 0, in_bytes + 64, &hi))) return rc;
    if ((rc = fx_pinned(e, 1, out_bytes + 64, &ho))) return rc;
    FX_HIP(e, hipHostGetDevicePointer(&di, hi, 0));
    FX_HIP(e, hipHostGetDevicePointer(&dout, ho, 0));
    *z = FxZeroCopy{(char*)hi, (char*)di, (char*)ho, (char*)dout};
    return FX_OK;
}

int fx_upload_lut(fx_engine* e, const uint8_t lut[256]) {
    if (e->lut_valid && std::memcmp(lut, e->h_lut, 256) == 0) return FX_OK;
    std::memcpy(e->h_lut, lut, 256);
    // h_lut lives in the engine: safe source for an async copy
    FX_HIP(e, hipMemcpyAsync(e->d_lut, e->h_lut, 256, hipMemcpyHostToDevice, e->stream));
    // a later call with a different LUT must not overwrite h_lut while this copy is pending
    FX_HIP(e, hipStreamSynchronize(e->stream));
    e->lut_valid = true;
    return FX_OK;
}

int fx_trace_buffer(fx_engine* e, unsigned long long** out) {
    *out = nullptr;
    if (!e->trace) return FX_OK;
#if !defined(FX_TRACE)
    return fx_fail(e, FX_EUNSUPPORTED, "this build has no in-kernel timeline: use the `make trace` build (FLEXS_AMD_LIB=.../libflexs_amd_trace.so)");
#endif
    if (!e->d_trace) {
        if (hipMalloc(reinterpret_cast<void**>(&e->d_trace), FX_TRACE_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            return fx_fail(e, FX_ENOMEM, "hipMalloc of the trace buffer failed");
        }
    }
    FX_HIP(e, hipMemsetAsync(e->d_trace, 0, FX_TRACE_BYTES, e->stream));
    *out = e->d_trace;
    return FX_OK;
}

static int check_deferred(fx_engine* e) {
    // caller has synchronised the stream; the error word lives in mapped pinned host memory,
    // so reading it costs nothing (no extra hipMemcpy on the small-call latency path)
    const unsigned err = *reinterpret_cast<volatile unsigned*>(e->h_err);
    if (err) {
        *reinterpret_cast<volatile unsigned*>(e->h_err) = 0;
        if (err & FX_ERR_TIMEOUT) {
            // (the barrier counter no longer matches what the host has added up: start over)
            if (e->d_lp_bar) { (void)hipMemset(e->d_lp_bar, 0, FX_LP_BAR_BYTES); for (unsigned& t : e->lp_bar_total) t = 0; }
            if (e->d_zero_pool) (void)hipMemset(e->d_zero_pool, 0, e->zero_pool_bytes);   // (partial maxima / tickets may be left behind)
            return fx_fail(e, FX_ESTATE, "a device-side barrier of the layer-parallel CNN form timed out (workgroups not co-resident?): set the engine option cnn_lp = 0");
        }
        if (err & FX_ERR_BADCHAR) return fx_fail(e, FX_EBADCHAR, "substring not found: character outside the alphabet");
    }
    return FX_OK;
}

extern "C" {

// ------------------------------------------------------------------ library
int fx_version(void) { return FX_VERSION; }

const char* fx_status_name(int s) {
    switch (s) {
        case FX_OK: return "FX_OK";
        case FX_EINVAL: return "FX_EINVAL";
        case FX_ESHAPE: return "FX_ESHAPE";
        case FX_EBADCHAR: return "FX_EBADCHAR";
        case FX_ENODEV: return "FX_ENODEV";
        case FX_EHIP: return "FX_EHIP";
        case FX_ENOMEM: return "FX_ENOMEM";
        case FX_EUNSUPPORTED: return "FX_EUNSUPPORTED";
        case FX_ESTATE: return "FX_ESTATE";
    }
    return "FX_UNKNOWN";
}

int fx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ------------------------------------------------------------------- engine
int fx_engine_create(int device, fx_engine** out) {
    if (!out) return FX_EINVAL;
    *out = nullptr;
    int n = fx_device_count();
    if (n <= 0 || device < 0 || device >= n) return fx_fail(nullptr, FX_ENODEV, "no such HIP device");
    fx_engine* e = new (std::nothrow) fx_engine();
    if (!e) return FX_ENOMEM;
    e->device = device;
#define FX_CREATE_HIP(call)                                                                   \
    do { hipError_t _r = (call); if (_r != hipSuccess) {                                      \
        fx_fail(nullptr, FX_EHIP, std::string(#call) + ": " + hipGetErrorString(_r)); delete e; return FX_EHIP; } } while (0)
    FX_CREATE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FX_CREATE_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // every kernel in this library is compiled for gfx950 only (MFMA f32 16x16x4, 160 KiB LDS, 8 XCDs)
        fx_fail(nullptr, FX_ENODEV, std::string("HIP device is ") + prop.gcnArchName + ", not gfx950 (MI355X)");
        delete e;
        return FX_ENODEV;
    }
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->large_bar = prop.isLargeBar != 0;
    e->max_lds = (int)std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, 64 * 1024);
    if (e->max_lds > 160 * 1024) e->max_lds = 160 * 1024;
    FX_CREATE_HIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    FX_CREATE_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < fx_engine::MAX_PIECES; ++i) {
        FX_CREATE_HIP(hipEventCreateWithFlags(&e->ev_in[i], hipEventDisableTiming));
        FX_CREATE_HIP(hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming));
        FX_CREATE_HIP(hipEventCreateWithFlags(&e->ev_out[i], hipEventDisableTiming));
    }
    FX_CREATE_HIP(hipEventCreate(&e->ev0));
    FX_CREATE_HIP(hipEventCreate(&e->ev1));
    FX_CREATE_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->h_err), 64, hipHostMallocMapped));
    *e->h_err = 0;
    FX_CREATE_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->h_done), 64, hipHostMallocMapped | hipHostMallocCoherent));
    *e->h_done = 0;
    FX_CREATE_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_done), e->h_done, 0));
    FX_CREATE_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_err), e->h_err, 0));
    FX_CREATE_HIP(hipMalloc(&e->d_lut, 256));
#undef FX_CREATE_HIP
    *out = e;
    return FX_OK;
}

int fx_engine_destroy(fx_engine* e) {
    if (!e) return FX_OK;
    (void)hipSetDevice(e->device);
    if (e->server.h_out) {
        e->server.running = e->server.in != nullptr;       // (whatever the bookkeeping says: tell them)
        fx_server_stop(e);
        for (hipStream_t st : e->server.streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        (void)hipHostFree((void*)e->server.h_out);
        if (e->server.in) (void)hipFree(e->server.in);
        e->server.h_out = nullptr;
    }
    fx_lp_disarm(e);
    (void)hipStreamSynchronize(e->stream);
    for (auto& p : e->d_scratch) if (p) (void)hipFree(p);
    if (e->d_zero_pool) (void)hipFree(e->d_zero_pool);
    if (e->d_train) (void)hipFree(e->d_train);
    if (e->h_train) (void)hipHostFree(e->h_train);
    if (e->d_train_dbg) (void)hipFree(e->d_train_dbg);
    for (auto& p : e->h_pinned) if (p) (void)hipHostFree(p);
    if (e->h_err) (void)hipHostFree(e->h_err);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->d_trace) (void)hipFree(e->d_trace);
    if (e->d_lp_bar) (void)hipFree(e->d_lp_bar);
    if (e->h_done) (void)hipHostFree(e->h_done);
    if (e->lp_mail) (void)hipFree(e->lp_mail);
    if (e->h_lp_state) (void)hipHostFree(e->h_lp_state);
    for (int i = 0; i < fx_engine::MAX_PIECES; ++i) {
        if (e->ev_in[i]) (void)hipEventDestroy(e->ev_in[i]);
        if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]);
        if (e->ev_out[i]) (void)hipEventDestroy(e->ev_out[i]);
    }
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return FX_OK;
}

int fx_engine_set_stream(fx_engine* e, void* hip_stream) {
    if (!e) return FX_EINVAL;
    FX_HIP(e, hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return FX_OK;
}

int fx_engine_sync(fx_engine* e) {
    if (!e) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

const char* fx_last_error(fx_engine* e) { return e ? e->last_error.c_str() : g_last_error_noengine.c_str(); }

static int64_t* option_slot(fx_engine* e, const char* key) {
    if (!e || !key) return nullptr;
    if (!std::strcmp(key, "force_generic")) return &e->force_generic;
    if (!std::strcmp(key, "cnn_variant")) return &e->cnn_variant;
    if (!std::strcmp(key, "grid_blocks")) return &e->grid_blocks;
    if (!std::strcmp(key, "cnn_conv1_mfma")) return &e->cnn_conv1_mfma;
    if (!std::strcmp(key, "mlp_l1_mfma")) return &e->mlp_l1_mfma;
    if (!std::strcmp(key, "mlp_pair")) return &e->mlp_pair;
    if (!std::strcmp(key, "stage_bytes")) return &e->stage_bytes;
    if (!std::strcmp(key, "stage_fill")) return &e->stage_fill;
    if (!std::strcmp(key, "dma_fill")) return &e->dma_fill;
    if (!std::strcmp(key, "cnn_pair_seg4")) return &e->cnn_pair_seg4;
    if (!std::strcmp(key, "cnn_seg_multi")) return &e->cnn_seg_multi;
    if (!std::strcmp(key, "dense_small")) return &e->dense_small;
    if (!std::strcmp(key, "cnn_quad")) return &e->cnn_quad;
    if (!std::strcmp(key, "cnn_lp")) return &e->cnn_lp;
    if (!std::strcmp(key, "cnn_lp_debug")) return &e->cnn_lp_debug;
    if (!std::strcmp(key, "cnn_pair")) return &e->cnn_pair;
    if (!std::strcmp(key, "cnn_pair_seg")) return &e->cnn_pair_seg;
    if (!std::strcmp(key, "cnn_seg")) return &e->cnn_seg;
    if (!std::strcmp(key, "dense_slab")) return &e->dense_slab;
    if (!std::strcmp(key, "cnn_big_units")) return &e->cnn_big_units;
    if (!std::strcmp(key, "poison_outputs")) return &e->poison_outputs;
    if (!std::strcmp(key, "trace")) return &e->trace;
    if (!std::strcmp(key, "ge_bytetab")) return &e->ge_bytetab;
    if (!std::strcmp(key, "wave_prio")) return &e->wave_prio;
    if (!std::strcmp(key, "dense_pipe")) return &e->dense_pipe;
    if (!std::strcmp(key, "fuse_mean")) return &e->fuse_mean;
    if (!std::strcmp(key, "dense_coop")) return &e->dense_coop;
    if (!std::strcmp(key, "quad_rotate")) return &e->quad_rotate;
    if (!std::strcmp(key, "serve_small")) return &e->serve_small;
    if (!std::strcmp(key, "serve_wide")) return &e->serve_wide;
    if (!std::strcmp(key, "serve_reserve_cus")) return &e->serve_reserve_cus;
    if (!std::strcmp(key, "serve_poll_sleep")) return &e->serve_poll_sleep;
    if (!std::strcmp(key, "serve_fence")) return &e->serve_fence;
    if (!std::strcmp(key, "serve_quads")) return &e->serve_quads;
    if (!std::strcmp(key, "serve_tiny")) return &e->serve_tiny;
    if (!std::strcmp(key, "dist_stage")) return &e->dist_stage;
    if (!std::strcmp(key, "dist_bounded")) return &e->dist_bounded;
    if (!std::strcmp(key, "host_mean_below")) return &e->host_mean_below;
    if (!std::strcmp(key, "done_flag")) return &e->done_flag;
    if (!std::strcmp(key, "lp_prelaunch")) return &e->lp_prelaunch;
    if (!std::strcmp(key, "serve_idle_us")) return &e->serve_idle_us;
    if (!std::strcmp(key, "chunk_overlap")) return &e->chunk_overlap;
    if (!std::strcmp(key, "zero_copy_bytes")) return &e->zero_copy_bytes;
    if (!std::strcmp(key, "zero_copy_mode")) return &e->zero_copy_mode;
    if (!std::strcmp(key, "train_rows")) return &e->train_rows;
    if (!std::strcmp(key, "train_lds")) return &e->train_lds;
    if (!std::strcmp(key, "train_threads")) return &e->train_threads;
    if (!std::strcmp(key, "train_trace")) return &e->train_trace;
    if (!std::strcmp(key, "train_persistent")) return &e->train_persistent;
    if (!std::strcmp(key, "train_canon")) return &e->train_canon;
    if (!std::strcmp(key, "train_swizzle")) return &e->train_swizzle;
    if (!std::strcmp(key, "train_split")) return &e->train_split;
    if (!std::strcmp(key, "dense_waves")) return &e->dense_waves;
    if (!std::strcmp(key, "dense_few_waves_below")) return &e->dense_few_waves_below;
    return nullptr;
}
// Kernel forms that were measured and lost (csrc/OPTIONS.md, "negative results") are compiled into the A/B build only
// (`make -C flexs_amd/csrc ab` -> libflexs_amd_ab.so, -DFX_AB; FLEXS_AMD_LIB selects it): the production library refuses
// the option values that would select them instead of silently running something else.
static void lp_disarm(fx_engine* e);
static bool ab_only_value(const fx_engine* e, const int64_t* s, int64_t value) {
#if defined(FX_AB)
    (void)e; (void)s; (void)value;
    return false;
#else
    if (s == &e->dense_pipe || s == &e->fuse_mean || s == &e->chunk_overlap || s == &e->cnn_conv1_mfma || s == &e->mlp_l1_mfma ||
        s == &e->dense_few_waves_below || s == &e->train_split)
        return value != 0;
    if (s == &e->dense_waves) return value == 8;
    if (s == &e->cnn_pair) return value == 0;
    if (s == &e->serve_quads) return value != 1;
    if (s == &e->cnn_variant) return value == 2 || value == 3 || value == 5 || value == 6;
    return false;
#endif
}

int fx_engine_set_option(fx_engine* e, const char* key, int64_t value) {
    int64_t* s = option_slot(e, key);
    if (!s) return fx_fail(e, FX_EINVAL, std::string("unknown option ") + (key ? key : "(null)"));
    if (ab_only_value(e, s, value))
        return fx_fail(e, FX_EUNSUPPORTED, std::string("option ") + key + " = " + std::to_string(value) + " selects a kernel form of the A/B build "
                       "(measured slower, see csrc/OPTIONS.md): make -C flexs_amd/csrc ab, FLEXS_AMD_LIB=.../libflexs_amd_ab.so");
    // a running resident generation was started under the old options (its geometry, but also the kernel forms its
    // workgroups run: pair rows or plain rows, ...): it leaves, the next calls start a new one under the new ones
    if (*s != value) { fx_server_stop(e); lp_disarm(e); }
    *s = value;
    e->server.refused.clear();                             // (what has a resident form depends on the form selectors)
    return FX_OK;
}
int fx_engine_get_option(fx_engine* e, const char* key, int64_t* value) {
    if (!value) return FX_EINVAL;
    if (e && key && !std::strcmp(key, "num_cus")) { *value = e->num_cus; return FX_OK; }
#if defined(FX_AB)
    if (e && key && !std::strcmp(key, "ab_build")) { *value = 1; return FX_OK; }
#else
    if (e && key && !std::strcmp(key, "ab_build")) { *value = 0; return FX_OK; }
#endif
    // read-only: the resident form's bookkeeping (requests answered, generations started, requests that fell back to a launch)
    if (e && key && !std::strcmp(key, "server_calls")) { *value = e->server.served; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_starts")) { *value = e->server.started; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_fallbacks")) { *value = e->server.fallbacks; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_last_fallback")) { *value = e->server.fb_info; return FX_OK; }   // reason (1 left, 2 timed out) | member | sequence | waited us
    if (e && key && !std::strcmp(key, "server_resident")) { *value = e->server.running ? 1 : 0; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_wide")) { *value = (e->server.running && e->server.wide) ? 1 : 0; return FX_OK; }
    if (e && key && !std::strncmp(key, "server_prof_", 12) && key[12] >= '0' && key[12] <= '7' && !key[13]) { *value = e->server.prof_ns[key[12] - '0']; return FX_OK; }
    if (e && key && !std::strncmp(key, "call_prof_", 10) && key[10] >= '0' && key[10] <= '3' && !key[11]) { *value = e->call_prof_ns[key[10] - '0']; return FX_OK; }
    if (e && key && !std::strncmp(key, "train_prof_", 11) && key[11] >= '0' && key[11] <= '4' && !key[12]) { *value = e->train_prof_ns[key[11] - '0']; return FX_OK; }
    if (e && key && !std::strcmp(key, "lp_armed_served")) { *value = e->lp_armed_served; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_streamed")) { *value = e->server.streamed; return FX_OK; }
    if (e && key && !std::strcmp(key, "server_slots")) { *value = e->server.running ? e->server.tiles : 0; return FX_OK; }
    int64_t* s = option_slot(e, key);
    if (!s) return fx_fail(e, FX_EINVAL, std::string("unknown option ") + (key ? key : "(null)"));
    *value = *s;
    return FX_OK;
}

int fx_engine_counters(fx_engine* e, int64_t* out9, int reset) {
    if (!e || !out9) return FX_EINVAL;
    const fx_engine::Counters& c = e->counters;
    const int64_t v[9] = {c.host_calls, c.device_calls, c.sequences, c.forwards, c.bytes_h2d, c.bytes_d2h, c.zero_copy_calls,
                          c.pair_evals, c.train_steps};
    std::memcpy(out9, v, sizeof(v));
    if (reset) e->counters = fx_engine::Counters{};
    return FX_OK;
}
int fx_timer_start(fx_engine* e) {
    if (!e) return FX_EINVAL;
    FX_HIP(e, hipEventRecord(e->ev0, e->stream));
    return FX_OK;
}
int fx_timer_stop(fx_engine* e, float* ms) {
    if (!e || !ms) return FX_EINVAL;
    FX_HIP(e, hipEventRecord(e->ev1, e->stream));
    FX_HIP(e, hipEventSynchronize(e->ev1));
    FX_HIP(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return FX_OK;
}

// -------------------------------------------------------------------- model
int fx_model_create(fx_engine* e, int kind, int L, int A, int F, int H, int K, fx_model** out) {
    if (!e || !out) return FX_EINVAL;
    *out = nullptr;
    if (kind < FX_CNN || kind > FX_GE) return fx_fail(e, FX_EINVAL, "unknown model kind");
    if (L < 1 || A < 1 || H < 1 || A > 254) return fx_fail(e, FX_EINVAL, "bad model dimensions");
    if (kind == FX_CNN) {
        if (F < 1 || K < 1 || A < 2) return fx_fail(e, FX_EINVAL, "bad CNN dimensions");
        // Keras 'valid' Conv1D raises at construction when seq_len < kernel_size (cnn.py:25-32)
        if (L < K) return fx_fail(e, FX_ESHAPE, "Negative dimension size: seq_len < kernel_size for 'valid' Conv1D");
    } else {
        F = 0; K = 0;
    }
    fx_model* m = new (std::nothrow) fx_model();
    if (!m) return FX_ENOMEM;
    m->eng = e;
    m->shape = FxShape{kind, L, A, F, H, K};
    m->layout = fx_pack_layout(m->shape);
    m->mfma_per_tile = fx_mfma_per_tile(m->shape);
    const int64_t np = fx_num_params(m->shape);
    m->blob.assign((size_t)np, 0.f);
    FX_HIP(e, hipSetDevice(e->device));
    if (hipMalloc(&m->d_blob, sizeof(float) * (size_t)np) != hipSuccess ||
        hipMalloc(&m->d_packed, sizeof(float) * (size_t)m->layout.alloc_floats) != hipSuccess) {
        (void)hipGetLastError();
        if (m->d_blob) (void)hipFree(m->d_blob);
        delete m;
        return fx_fail(e, FX_ENOMEM, "hipMalloc of model weights failed");
    }
    *out = m;
    return FX_OK;
}

int fx_model_destroy(fx_model* m) {
    if (!m) return FX_OK;
    (void)hipSetDevice(m->eng->device);
    {
        // resident workgroups may read this model's weights: tell them to leave, wait for them; forget what was remembered about it
        auto& sv = m->eng->server;
        if (std::find(sv.models.begin(), sv.models.end(), m) != sv.models.end()) {
            fx_server_stop(m->eng);
            for (int g = 0; g < sv.groups; ++g) (void)hipStreamSynchronize(sv.streams[g]);
            sv.models.clear();
        }
        sv.pending.clear();
        sv.refused.clear();
    }
    (void)hipStreamSynchronize(m->eng->stream);
    if (m->d_blob) (void)hipFree(m->d_blob);
    if (m->d_packed) (void)hipFree(m->d_packed);
    if (m->d_bytetab) (void)hipFree(m->d_bytetab);
    delete m;
    return FX_OK;
}

int64_t fx_model_num_params(const fx_model* m) { return m ? (int64_t)m->blob.size() : FX_EINVAL; }

int fx_model_set_weights(fx_model* m, const float* blob, int64_t n) {
    if (!m || !blob) return FX_EINVAL;
    fx_engine* e = m->eng;
    if (n != (int64_t)m->blob.size()) return fx_fail(e, FX_ESHAPE, "weight blob has the wrong number of floats");
    FX_HIP(e, hipSetDevice(e->device));
    std::memcpy(m->blob.data(), blob, sizeof(float) * (size_t)n);
    std::vector<float> packed((size_t)m->layout.alloc_floats);
    fx_pack_weights(m->shape, m->blob.data(), packed.data());
    // in-flight kernels may still read the old weights (a resident generation holds them in LDS: it is told to leave,
    // and the version makes the next call start a new one)
    m->version += 1;
    if (e->server.running) {
        fx_server_stop(e);
        for (int g = 0; g < e->server.groups; ++g) FX_HIP(e, hipStreamSynchronize(e->server.streams[g]));
    }
    FX_HIP(e, hipStreamSynchronize(e->stream));
    FX_HIP(e, hipMemcpy(m->d_blob, m->blob.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    FX_HIP(e, hipMemcpy(m->d_packed, packed.data(), sizeof(float) * packed.size(), hipMemcpyHostToDevice));
    m->has_weights = true;
    m->bt_valid = false;
    return FX_OK;
}

int fx_model_get_weights(const fx_model* m, float* blob, int64_t n) {
    if (!m || !blob) return FX_EINVAL;
    if (n != (int64_t)m->blob.size()) return FX_ESHAPE;
    std::memcpy(blob, m->blob.data(), sizeof(float) * (size_t)n);
    return FX_OK;
}

// ------------------------------------------------------------------ scoring
// Plane stride of the engine's member-major intermediate for N sequences (floats; 256-byte aligned planes).
static inline int64_t planar_stride_for(int64_t N) { return (N + 63) & ~(int64_t)63; }

// planar_stride == 0: d_NM is the row-major (N, M) matrix of the ABI; > 0: M member planes that far apart.
static void server_stop(fx_engine* e);
static void lp_disarm(fx_engine* e);
static int score_dispatch(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                          float* d_NM, int64_t planar_stride = 0) {
    if (!e->lp_arm_next) lp_disarm(e);                     // (a pre-launched instance of another call shape holds the CUs: it leaves)
    e->lp_launches = 0;
    e->dispatch_groups = 0;
    // a launch whose workgroups would not all find a CU beside the resident ones tells those to leave (they hold most of their
    // CU's LDS: a persistent workgroup that has to wait for one of them would wait for their idle exit)
    if (e->server.running && (int64_t)M * ((N + 15) / 16) + e->server.wgs > e->num_cus) server_stop(e);
    if (e->poison_outputs && !e->lp_arm_next)   // scores are nan_to_num'ed, so a NaN that survives is an element no kernel wrote
        // (not for a pre-launched instance: the memset would run between the end of the instance being answered and the host
        //  reading ITS results from the same pinned planes; lp_serve_armed poisons them on the host instead)
        FX_HIP(e, hipMemsetAsync(d_NM, 0xFF, sizeof(float) * (planar_stride ? (size_t)planar_stride * (size_t)M : (size_t)N * (size_t)M), e->stream));
    struct Layout {                                     // the launchers read the layout from the engine
        fx_engine* e;
        Layout(fx_engine* e_, int64_t s) : e(e_) { e->planar_stride = s; }
        ~Layout() { e->planar_stride = 0; }
    } layout(e, planar_stride);
    // group consecutive members into launches of <= FX_MAX_M homogeneous models
    for (int m0 = 0; m0 < M;) {
        int cnt = 1;
        const FxShape& s0 = models[m0]->shape;
        while (m0 + cnt < M && cnt < FX_MAX_M) {
            const FxShape& s = models[m0 + cnt]->shape;
            if (s.kind != s0.kind || s.F != s0.F || s.H != s0.H || s.K != s0.K) break;
            ++cnt;
        }
        int rc = FX_EUNSUPPORTED;
        e->dispatch_groups += 1;
        e->done_armed = false;                             // (only the LAST launch of a dispatch may offer the completion flag)
        if (!e->force_generic) {
            if (s0.kind == FX_CNN) {
                rc = fx_launch_score_cnn_mfma(e, models + m0, cnt, d_ascii, N, d_NM, M, m0);
                if (rc == FX_EUNSUPPORTED) rc = fx_launch_score_cnn_split(e, models + m0, cnt, d_ascii, N, d_NM, M, m0);
            }
            else rc = fx_launch_score_dense_mfma(e, models + m0, cnt, d_ascii, N, d_NM, M, m0);
        }
        if (rc == FX_EUNSUPPORTED) rc = fx_launch_score_generic(e, models + m0, cnt, d_ascii, N, d_NM, M, m0);
        if (rc) return rc;
        m0 += cnt;
    }
    (void)L;
    return FX_OK;
}

// score into member-major planes, then the NumPy-order mean -- in the scoring kernel itself where a launcher offers it
static int score_then_mean(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                           float* d_planes, int64_t stride, float* mean_dst) {
    e->fuse_mean_out = (e->fuse_mean && stride && M > 1 && M <= 16) ? mean_dst : nullptr;
    e->fused_mean_done = false;
    const int rc = score_dispatch(e, models, M, d_ascii, N, L, d_planes, stride);
    e->fuse_mean_out = nullptr;
    if (rc) return rc;
    if (e->fused_mean_done) return FX_OK;
    return fx_launch_ensemble_mean_planar(e, d_planes, N, M, stride, mean_dst);
}

static int validate_models(fx_engine* e, fx_model* const* models, int M, int L, const uint8_t* lut) {
    if (!e || !models || M < 1 || !lut) return FX_EINVAL;
    for (int m = 0; m < M; ++m) {
        if (!models[m]) return fx_fail(e, FX_EINVAL, "null model handle");
        if (models[m]->eng != e) return fx_fail(e, FX_EINVAL, "model belongs to another engine");
        if (!models[m]->has_weights) return fx_fail(e, FX_ESTATE, "model weights were never set");
        if (models[m]->shape.L != L) return fx_fail(e, FX_ESHAPE, "sequence length does not match the model's seq_len");
        if (models[m]->shape.A != models[0]->shape.A) return fx_fail(e, FX_ESHAPE, "ensemble members use different alphabets");
    }
    for (int c = 0; c < 256; ++c)
        if (lut[c] != 0xFF && lut[c] >= models[0]->shape.A) return fx_fail(e, FX_EINVAL, "LUT entry >= alphabet size");
    return FX_OK;
}

int fx_score_dev(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                 const uint8_t lut[256], float* d_out_NM, float* d_out_mean) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 0) return fx_fail(e, FX_EINVAL, "negative batch size");
    if (N == 0) return FX_OK;
    if (!d_ascii || (!d_out_NM && !d_out_mean)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    e->counters.device_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    rc = fx_upload_lut(e, lut);
    if (rc) return rc;
    float* d_NM = d_out_NM;
    if (!d_NM) {
        // only the mean is wanted: the intermediate is the engine's own, laid out member-major (contiguous stores)
        const int64_t stride = M <= 16 ? planar_stride_for(N) : 0;
        void* p = nullptr;
        rc = fx_scratch(e, 1, sizeof(float) * (stride ? (size_t)stride * (size_t)M : (size_t)N * (size_t)M), &p);
        if (rc) return rc;
        d_NM = (float*)p;
        if (stride) return score_then_mean(e, models, M, d_ascii, N, L, d_NM, stride, d_out_mean);
    }
    rc = score_dispatch(e, models, M, d_ascii, N, L, d_NM);
    if (rc) return rc;
    if (d_out_mean) rc = fx_launch_ensemble_reduce(e, d_NM, N, M, nullptr, d_out_mean, nullptr);
    return rc;
}

// The two halves of the mean-only device path, for callers that keep the intermediate themselves (bench.py times
// them separately): scores as M member-major planes `stride` floats apart, then np.mean over the planes.
int fx_score_planes_dev(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                        const uint8_t lut[256], float* d_planes, int64_t stride) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 0 || stride < N || (stride & 3)) return fx_fail(e, FX_EINVAL, "fx_score_planes_dev: stride must be >= N and a multiple of 4");
    if (N == 0) return FX_OK;
    if (!d_ascii || !d_planes || (reinterpret_cast<uintptr_t>(d_planes) & 15)) return fx_fail(e, FX_EINVAL, "null or unaligned buffer");
    FX_HIP(e, hipSetDevice(e->device));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    e->counters.device_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    return score_dispatch(e, models, M, d_ascii, N, L, d_planes, stride);
}

int fx_ensemble_mean_planes_dev(fx_engine* e, const float* d_planes, int64_t N, int M, int64_t stride, float* d_out_mean) {
    if (!e || N < 0 || M < 1 || M > 16 || stride < N || (stride & 3)) return FX_EINVAL;
    if (N == 0) return FX_OK;
    if (!d_planes || !d_out_mean || (reinterpret_cast<uintptr_t>(d_planes) & 15)) return fx_fail(e, FX_EINVAL, "null or unaligned buffer");
    FX_HIP(e, hipSetDevice(e->device));
    return fx_launch_ensemble_mean_planar(e, d_planes, N, M, stride, d_out_mean);
}

int fx_staging_input(fx_engine* e, int64_t bytes, void** host) {
    if (!e || bytes < 0 || !host) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    return fx_pinned(e, 0, (size_t)std::max<int64_t>(bytes, 1), host);
}

// How a host call (host bytes in, host scores out) should move its data.
//   zero-copy  the kernels read the sequences from the mapped pinned staging area over PCIe and write the scores to
//              pinned memory: no copy enqueues, no dependent copy -> kernel -> copy chain.  Every MEMBER's units read the
//              bytes again (host memory is not cached in L2), so the traffic is M x N x L bytes: worth it when that
//              hides behind the kernels (3 x CNN L = 8: 24 bytes per sequence against 0.3 MFLOP), ruinous when it does
//              not (8 x GlobalEpistasis L = 90: 72 MB for a 0.15 ms launch; measured 1.43 ms vs 0.44 ms).
//   pieces     > 1: pack + submit in pieces so that the host's string marshalling overlaps the GPU's work.
// Model: t_kernel from the MFMA instructions the launch issues (fx_mfma_per_tile) at 75 % of the pipe; PCIe at 45 GB/s
// for in-kernel reads, 35 GB/s + 25 us of enqueue / dependency latency for the copy path (profiles/r3_e2e_ab.log).
static void plan_host_call(const fx_engine* e, fx_model* const* models, int M, int64_t N, int L, bool* zero_copy, int* pieces) {
    const double bytes = (double)N * (double)L;
    double t_k = 0.0;
    bool mfma = true;
    for (int m = 0; m < M; ++m) {
        const int64_t per_tile = models[m]->mfma_per_tile;       // (3.7 us per call of a three-member 237-residue ensemble when recomputed here)
        if (per_tile < 0) { mfma = false; break; }
        t_k += (double)per_tile * (double)((N + 15) / 16) * 32.0 / ((double)e->num_cus * 4.0 * 2.4e9) / 0.75;
    }
    const double t_zc = std::max(t_k, (double)M * bytes / 45e9);
    const double t_copy = bytes / 35e9 + t_k + 25e-6;
    bool zc = mfma && t_zc < t_copy;
    if (e->zero_copy_mode == 0) zc = false;
    if (e->zero_copy_mode == 1) zc = true;
    // Pieces only pay when the host's marshalling (~20 GB/s with the packing threads + ~1 ns per string) is a visible
    // share of the call AND every piece still fills the machine for a while (a piece shorter than ~0.25 ms of kernel
    // time loses more to its start-up and tail than the overlap wins: 7 pieces of a 17.6 ms protein batch cost 3 ms).
    const double t_pack = bytes / 20e9 + (double)N * 1e-9;
    int p = 1;
    if (t_pack > 0.15 * t_k || !mfma) {
        p = zc ? (int)(bytes / (2 << 20) + 0.5)                           // ~2 MB of sequence bytes per piece
               : (bytes >= (double)(16 << 20) ? (int)(bytes / (4 << 20)) : 1);   // big uploads: 4 MB pieces
        const int cap = mfma ? (int)(t_k / 250e-6) : 16;
        if (p > cap) p = cap;
    }
    *zero_copy = zc;
    *pieces = p < 1 ? 1 : (p > 16 ? 16 : p);
}

int fx_plan_host_call(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, int* zero_copy, int* pieces) {
    if (!e || !models || M < 1 || N < 0 || L < 1 || !zero_copy || !pieces) return FX_EINVAL;
    for (int m = 0; m < M; ++m) if (!models[m]) return FX_EINVAL;
    bool zc = false;
    plan_host_call(e, models, M, N, L, &zc, pieces);
    *zero_copy = zc ? 1 : 0;
    return FX_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Resident small-call form: the host side of the mailboxes (score_cnn_quad.hip, SERVER; FxMailIn / FxMailOut).
static void server_stop(fx_engine* e) { fx_server_stop(e); }

// Can the host store into this device allocation?  The device reports a large BAR, but whether THIS allocation is mapped
// into the process is the runtime's business.  Decided from facts, without ever faulting (round 3 probed with a guarded store
// under a temporary SIGSEGV handler: not thread-safe, and hostile to a host application that owns its signal handlers):
//   1. the address range must be a readable + writable mapping of this process (/proc/self/maps);
//   2. a magic word stored through that mapping must be what a device -> host copy of the same address returns.
// Serialised by a mutex (engines on different threads), decided once per allocation.
static std::mutex g_probe_mu;
static bool host_range_is_writable(const void* p, size_t len) {
    FILE* f = std::fopen("/proc/self/maps", "r");
    if (!f) return false;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + len;
    uintptr_t covered = lo;                               // the range may span adjacent mappings (listed in address order)
    char line[512];
    bool ok = false;
    while (std::fgets(line, sizeof line, f)) {
        unsigned long long a = 0, b = 0;
        char perm[8] = {};
        if (std::sscanf(line, "%llx-%llx %7s", &a, &b, perm) != 3) continue;
        if (b <= covered) continue;
        if (a > covered) break;                            // a hole before the range is covered
        if (perm[0] != 'r' || perm[1] != 'w') break;
        covered = (uintptr_t)b;
        if (covered >= hi) { ok = true; break; }
    }
    std::fclose(f);
    return ok;
}
static bool host_can_store(FxMailIn* q) {
    std::lock_guard<std::mutex> lock(g_probe_mu);
    if (!host_range_is_writable(q, sizeof(FxMailIn))) return false;
    volatile unsigned* p = &q->stop;
    const unsigned magic = 0x5EB1A5EDu;
    *p = magic;
    fx_bar_fence();
    unsigned back = 0;
    if (hipMemcpy(&back, const_cast<const unsigned*>(p), sizeof back, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    *p = 0u;
    fx_bar_fence();
    return back == magic;
}

extern "C" int64_t fx_collect_lines(const volatile unsigned long long* ans, float* scores, int64_t n0, int64_t N, unsigned seq, int64_t ahead, int* bad);   // host_collect.cc

static int server_start(fx_engine* e, fx_model* const* models, int M, int L, const uint8_t lut[256]) {
    auto& sv = e->server;
    if (!e->large_bar) return FX_EUNSUPPORTED;             // the host must be able to store into device memory
    if (!sv.h_out) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(FxMailOut), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return FX_ENOMEM; }
        FxMailOut* d = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d), p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return FX_EHIP; }
        sv.h_out = new (p) FxMailOut();
        sv.d_out = d;
    }
    if (!sv.in) {
        FxMailIn* q = nullptr;
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&q), sizeof(FxMailIn), hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return FX_ENOMEM; }
        if (!host_can_store(q)) { (void)hipFree(q); e->large_bar = false; return FX_EUNSUPPORTED; }
        sv.in = q;
    }
    for (int g = 0; g < sv.groups; ++g) FX_HIP(e, hipStreamSynchronize(sv.streams[g]));   // a previous generation has left (it was told to, or timed out)
    std::memset((void*)sv.h_out->alive, 0, sizeof(sv.h_out->alive));   // (answers carry sequence numbers that never repeat: no need to clear them)
    sv.in->req = 0; sv.in->req_tail = 0; sv.in->req_wide = 0; sv.in->ready = 0; sv.in->stop = 0;  // (through the BAR, like every host access to it; posted before the launch's doorbell)
    fx_bar_fence();
    int rc = fx_upload_lut(e, lut);
    if (rc) return rc;
    FX_HIP(e, hipStreamSynchronize(e->stream));            // the LUT (and any weight upload) must have landed before the workgroups read them
    // one workgroup per (member, 16-sequence tile slot).  Wide generation (round 4): most of the chip -- slot s of a member
    // walks the tiles s, s + tiles, s + 2 tiles, ... of a request, so the capacity is what the mailboxes hold, not the slot
    // count; round 3's geometry (serve_wide = 0): a third of the chip, <= 16 slots, one tile per slot.
    // serve_wide = 1 (default): ADAPTIVE -- a generation is wide when the caller has recently asked for more than 256 sequences at
    // a time (server_call keeps the time of the last such request), narrow otherwise: 240 resident workgroups cost every
    // explorer-size call ~1.2 us (13.2 vs 12.0 us per 20-sequence call, profiles/r4_server_wide_ab.log) whatever they poll and
    // however long they sleep, and the callers that only ever ask for 1-20 sequences (Adalead's roll-outs, CMA-ES, DyNA-PPO)
    // should not pay it.  serve_wide = 2: always wide (A/B, tests); 0: round 3's geometry.
    const bool want_wide = e->serve_wide == 2 ||
        (e->serve_wide == 1 && sv.mid_recent > 0 &&
         std::chrono::duration<double>(std::chrono::steady_clock::now() - sv.t_mid).count() < 0.25);
    int tiles, cap;
    if (want_wide) {
        int reserve = (int)e->serve_reserve_cus;
        if (reserve < 0) reserve = 0;
        if (reserve > e->num_cus - M) reserve = e->num_cus - M;
        tiles = (e->num_cus - reserve) / M;
        if (tiles > FX_SERVE_TILES) tiles = FX_SERVE_TILES;
        if (tiles < 1) return FX_EUNSUPPORTED;
        cap = FX_SERVE_BYTES / L < FX_SERVE_CAP ? FX_SERVE_BYTES / L : FX_SERVE_CAP;
        // the first slots of every member -- as many as round 3's whole generation had -- spin on `req`; the rest poll `req_wide`
        sv.fast = e->num_cus / 3 / M;
        if (sv.fast > FX_SERVE_FAST) sv.fast = FX_SERVE_FAST;
        if (sv.fast < 1) sv.fast = 1;
        if (sv.fast > tiles) sv.fast = tiles;
    } else {
        tiles = e->num_cus / 3 / M;
        if (tiles > 16) tiles = 16;
        if (tiles > 16384 / (16 * L)) tiles = 16384 / (16 * L);
        if (tiles < 1) return FX_EUNSUPPORTED;
        cap = 16 * tiles;
        sv.fast = tiles;
    }
    // a host that stops asking (or dies) frees the CUs by itself: after 2 x serve_idle_us (100 MHz ticks), and 10 s whatever happens
    const unsigned long long idle = (unsigned long long)e->serve_idle_us * 200ull, life = 1000000000ull;
    // consecutive like members form a group; every group is its own resident launch on its own stream (all read the same
    // request and answer into their members' rows), so a mixed ensemble -- DyNA-PPO's GE + MLP + CNN -- is served too
    auto group_len = [&](int m0) {
        int cnt = 1;
        const FxShape& s0 = models[m0]->shape;
        while (m0 + cnt < M) {
            const FxShape& s = models[m0 + cnt]->shape;
            if (s.kind != s0.kind || s.F != s0.F || s.H != s0.H || s.K != s0.K) break;
            ++cnt;
        }
        return cnt;
    };
    // streams first (creating one takes milliseconds the first time; a group already launched would idle out meanwhile)
    int want_streams = 0;
    for (int m0 = 0; m0 < M; m0 += group_len(m0)) ++want_streams;
    while ((int)sv.streams.size() < want_streams) {
        // highest priority: the runtime keeps separate hardware queues per priority level, so a resident launch does not
        // sit in front of work that normal-priority streams (the engine's, PyTorch's) submit to a shared hardware queue
        hipStream_t st = nullptr;
        int prio_low = 0, prio_high = 0;
        FX_HIP(e, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        FX_HIP(e, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_high));
        sv.streams.push_back(st);
    }
    sv.groups = 0;
    for (int m0 = 0; m0 < M;) {
        const int cnt = group_len(m0);
        hipStream_t st = sv.streams[sv.groups];
        sv.groups += 1;
        int quads = 1;
        rc = fx_launch_score_cnn_quad_server(e, models + m0, cnt, m0, tiles, st, sv.in, sv.d_out, idle, life, want_wide ? (int)e->serve_quads : 1, &quads);
        for (int m = m0; m < m0 + cnt; ++m) sv.quads[m] = rc == FX_OK ? quads : 1;
        if (rc == FX_EUNSUPPORTED) rc = fx_launch_score_dense_small_server(e, models + m0, cnt, m0, tiles, st, sv.in, sv.d_out, idle, life);
        if (rc) {                                          // a member without a resident form: the groups already started leave again
            sv.in->stop = 1; sv.in->req_wide = FX_SERVE_LEAVE; sv.in->req = FX_SERVE_LEAVE;
            fx_bar_fence();
            return rc;
        }
        m0 += cnt;
    }
    sv.models.assign(models, models + M);
    sv.versions.clear();
    for (int m = 0; m < M; ++m) sv.versions.push_back(models[m]->version);
    std::memcpy(sv.lut, lut, 256);
    sv.L = L; sv.cap = cap; sv.wgs = M * tiles; sv.tiles = tiles; sv.wide = want_wide;
    sv.seen.assign((size_t)FX_MAX_M * FX_SERVE_TILES, 0);
    sv.running = true; sv.fresh = true;
    sv.t_start = std::chrono::steady_clock::now();
    sv.started += 1;
    return FX_OK;
}

// FX_OK: answered.  FX_EUNSUPPORTED: not this time (the caller launches as usual).  Anything else: the call's error.
// np.mean over the members (ensemble.py:24) of M member planes `stride` floats apart, on the host, M <= 16, in NumPy's order.
static void host_mean_planes(const float* pl, int64_t stride, int64_t N, int M, float* out_mean) {
    if (M < 8) {
        // NumPy's order for fewer than eight members is the plain left-to-right sum from 0 (np_sum_row): plane by plane,
        // which the compiler vectorises over the sequences
        for (int64_t n = 0; n < N; ++n) out_mean[n] = 0.f;
        for (int m = 0; m < M; ++m) {
            const float* pm = pl + (size_t)m * (size_t)stride;
            for (int64_t n = 0; n < N; ++n) out_mean[n] += pm[n];
        }
        const float fm = (float)M;
        for (int64_t n = 0; n < N; ++n) out_mean[n] = out_mean[n] / fm;
    } else {
        for (int64_t n = 0; n < N; ++n) {
            float x16[16];
            for (int m = 0; m < 16; ++m) x16[m] = m < M ? pl[(size_t)m * (size_t)stride + n] : 0.f;
            out_mean[n] = np_mean_row16(x16, M);           // NumPy's order, the same routine the mean kernels use
        }
    }
}

// ---- pre-launched instance of the layer-parallel protein form (fx_common.h LpArmed) -------------------------------------------
static bool lp_mail_ensure(fx_engine* e) {
    if (e->lp_mail) return true;
    if (e->lp_mail_refused || !e->large_bar) return false;
    e->lp_mail_refused = true;                             // (until proven otherwise: asked once)
    FxLpMail* q = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&q), sizeof(FxLpMail), hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return false; }
    {
        // can the host store into it?  (as for the resident form's mailbox: a writable mapping of this process + a read-back)
        std::lock_guard<std::mutex> lock(g_probe_mu);
        bool ok = host_range_is_writable(q, sizeof(FxLpMail));
        if (ok) {
            volatile unsigned long long* p = &q->req[0].w;
            *p = 0x5EB1A5ED5EB1A5EDull;
            fx_bar_fence();
            unsigned long long back = 0;
            ok = hipMemcpy(&back, const_cast<const unsigned long long*>(p), sizeof back, hipMemcpyDeviceToHost) == hipSuccess && back == 0x5EB1A5ED5EB1A5EDull;
            if (!ok) (void)hipGetLastError();
        }
        if (!ok) { (void)hipFree(q); return false; }
    }
    if (!e->h_lp_state) {
        if (hipHostMalloc(reinterpret_cast<void**>(&e->h_lp_state), 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_lp_state), e->h_lp_state, 0) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(q); return false;
        }
        *e->h_lp_state = 0;
    }
    for (auto& w : q->req) w.w = 0;
    fx_bar_fence();
    e->lp_mail = q;
    e->lp_mail_refused = false;
    return true;
}

// Tell a pre-launched instance to leave (harmless when it has left by itself) and put the layer-parallel form's barrier counters
// back: the host counted the instance's arrivals when it enqueued it.  Stream-ordered: the memset runs after the instance.
static void lp_disarm(fx_engine* e) {
    if (!e->lp_armed.on) return;
    e->lp_armed.on = false;
    for (auto& w : e->lp_mail->req) w.w = ((unsigned long long)e->lp_armed.seq << 16) | 0xFFFFull;
    fx_bar_fence();
    if (e->d_lp_bar) { (void)hipMemsetAsync(e->d_lp_bar, 0, FX_LP_BAR_BYTES, e->stream); for (unsigned& t : e->lp_bar_total) t = 0; }
}

}  // extern "C"
void fx_lp_disarm(fx_engine* e) { lp_disarm(e); }
extern "C" {

// Enqueue the NEXT instance of the call that was just answered by the layer-parallel form: same members, same batch size, results
// to the same pinned planes / matrix.  It fills its weights and waits for its request word.
static void lp_arm(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], float* out_dev, int64_t stride, int mode) {
    if (!e->lp_prelaunch || !e->done_flag || e->trace || !lp_mail_ensure(e)) return;
    e->lp_arm_next = true;
    const int rc = score_dispatch(e, models, M, e->lp_mail->bytes, N, L, out_dev, stride);
    e->lp_arm_next = false;
    e->done_armed = false;                                 // (nobody waits for this launch's flag until it has been asked)
    if (rc != FX_OK || e->lp_launches != 1 || e->dispatch_groups != 1) {
        // (not the layer-parallel form after all: whatever was enqueued scored the mailbox's old bytes into the scratch planes --
        //  harmless, and stream-ordered before anything that reads them)
        (void)hipGetLastError();
        return;
    }
    auto& a = e->lp_armed;
    a.on = true;
    a.models.assign(models, models + M);
    a.versions.clear();
    for (int m = 0; m < M; ++m) a.versions.push_back(models[m]->version);
    a.N = N; a.L = L; a.mode = mode; a.stride = stride;
    std::memcpy(a.lut, lut, 256);
    a.seq = e->done_seq;
    a.t = std::chrono::steady_clock::now();
}

// Is the pre-launched instance the one this call can use?
static bool lp_armed_matches(const fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], int64_t stride, int mode) {
    const auto& a = e->lp_armed;
    if (!a.on || (int)a.models.size() != M || a.N != N || a.L != L || a.mode != mode || a.stride != stride || std::memcmp(a.lut, lut, 256) != 0) return false;
    for (int m = 0; m < M; ++m) if (a.models[m] != models[m] || a.versions[m] != models[m]->version) return false;
    // (the instance leaves serve_idle_us after it started waiting: one that is about to is not asked any more)
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - a.t).count() * 1e6 < 0.6 * (double)e->serve_idle_us;
}

// The instance that was asked did not answer (it had left): it never arrived at the barrier counters the host counted forward for
// it, and a next instance that was enqueued behind it builds on those totals -- that one is told to leave, the counters go back to
// zero (stream-ordered behind both), whether or not a next instance was armed.
static void lp_unserved(fx_engine* e) {
    if (e->lp_armed.on) { lp_disarm(e); return; }
    if (e->d_lp_bar) { (void)hipMemsetAsync(e->d_lp_bar, 0, FX_LP_BAR_BYTES, e->stream); for (unsigned& t : e->lp_bar_total) t = 0; }
}

// Answer the call with the pre-launched instance: its sequences and request word go into the mailbox, the NEXT instance is
// enqueued while this one computes, then the completion flag.  false: the instance had left (the caller launches as usual).
static bool lp_serve_armed(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256],
                           float* out_dev, int64_t stride, int mode, void* out_host, size_t out_bytes) {
    auto& a = e->lp_armed;
    const unsigned seq = a.seq;
    if (e->poison_outputs) std::memset(out_host, 0xFF, out_bytes);     // (the instance writes pinned HOST memory: poisoned here)
    std::memcpy(e->lp_mail->bytes, ascii, (size_t)N * L);
    fx_bar_fence();
    const unsigned long long word = ((unsigned long long)seq << 16) | (unsigned long long)N;
    for (auto& w : e->lp_mail->req) w.w = word;
    fx_bar_fence();
    a.on = false;                                          // (being served: not to be cancelled by the dispatch below)
    // the barrier totals of the instance being served are part of what the next one builds on: enqueue it now, it starts when
    // this one is through
    const auto t0 = std::chrono::steady_clock::now();
    const volatile unsigned* done = e->h_done;
    const volatile unsigned* state = e->h_lp_state;
    const unsigned gone = (seq << 1) | 1u;
    bool armed_next = false;
    for (unsigned spins = 0;; ++spins) {
        if (*done == seq) break;
        if (*state == gone) {                              // it left just before the request arrived
            lp_unserved(e);
            return false;
        }
        if (!armed_next) {                                 // (after the first look: ~2.6 us of enqueue beside the instance's ~25 us of work)
            lp_arm(e, models, M, N, L, lut, out_dev, stride, mode);
            armed_next = true;
            continue;
        }
        __builtin_ia32_pause();
        if ((spins & 4095u) == 4095u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            (void)hipStreamSynchronize(e->stream);
            if (*done == seq) break;
            lp_unserved(e);
            return false;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    e->lp_armed_served += 1;
    return true;
}

// Wait for the results of the launches just enqueued on the engine's stream: the completion flag of the last launch where it
// offers one (poll: the word is in pinned host memory), else -- or when the flag stays away for 2 s -- the stream itself.
static int wait_for_results(fx_engine* e, unsigned want_seq = 0) {
    if ((e->done_armed || want_seq) && e->done_flag) {
        const volatile unsigned* w = e->h_done;
        const unsigned want = want_seq ? want_seq : e->done_seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; *w != want; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 4095u) == 4095u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
                FX_HIP(e, hipStreamSynchronize(e->stream));   // (a kernel that left through an error path never raises the flag)
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else if (e->done_flag && e->d_done) {
        // no kernel-side flag: a value written by the command processor behind the launches, polled in pinned host memory --
        // 8.4 us of wait beyond the kernels instead of hipStreamSynchronize's 11.1 (profiles/r4_sync_latency_probe.log)
        const unsigned v = ++e->done_value ? e->done_value : ++e->done_value;
        if (hipStreamWriteValue32(e->stream, e->d_done + 8, v, 0) != hipSuccess) {
            (void)hipGetLastError();
            FX_HIP(e, hipStreamSynchronize(e->stream));
        } else {
            const volatile unsigned* w = e->h_done + 8;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; *w != v; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 4095u) == 4095u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
                    FX_HIP(e, hipStreamSynchronize(e->stream));
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
    } else {
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    e->done_armed = false;
    return FX_OK;
}

// The explorer-size (zero-copy) forms of the small entry points -- distances, neighbour search, table look-ups, the fused
// NoisyAbstractModel query, the population step: everything they enqueued has finished when this returns.
static int fx_wait_small(fx_engine* e) {
    e->done_armed = false;
    return wait_for_results(e);
}

static int64_t server_since(const fx_engine* e) {
    return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - e->server.t_entry).count();
}

// May this request go to the resident form?  FX_OK: a generation of exactly these members is running (started here if the
// calls come densely enough) and holds N sequences.  FX_EUNSUPPORTED: not this time.
static int server_admit(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256]) {
    auto& sv = e->server;
    sv.t_entry = std::chrono::steady_clock::now();
    auto since = [&]() { return server_since(e); };
    if (sv.streaming) {                                    // a streamed call that was never closed (fx_score_stream_end): abandoned
        sv.streaming = false;
        server_stop(e);
    }
    if (!e->serve_small || e->trace || e->force_generic || N < 1 || N > FX_SERVE_CAP || N * L > FX_SERVE_BYTES || M > FX_MAX_M) return FX_EUNSUPPORTED;
    bool same = sv.running && (int)sv.models.size() == M && sv.L == L && std::memcmp(sv.lut, lut, 256) == 0;
    for (int m = 0; same && m < M; ++m) same = sv.models[m] == models[m] && sv.versions[m] == models[m]->version;
    if (sv.running && !same) server_stop(e);
    if (!sv.running && (int)sv.refused.size() == M && sv.refused_L == L && std::equal(sv.refused.begin(), sv.refused.end(), models))
        return FX_EUNSUPPORTED;                            // (an ensemble with a member that has no resident form: asked once)
    if (N > 256) {
        // adaptive geometry: remember when the caller last asked for more than a narrow generation holds; the second such
        // request within 2 ms while a narrow generation runs replaces it by a wide one (this call is launched)
        const auto now = std::chrono::steady_clock::now();
        const bool dense = sv.mid_recent > 0 && std::chrono::duration<double>(now - sv.t_mid).count() < 2e-3;
        sv.t_mid = now;
        sv.mid_recent = 1;
        if (e->serve_wide == 1 && sv.running && !sv.wide && dense) {
            server_stop(e);
            sv.pending.assign(models, models + M);         // (the next call of this ensemble within the window starts the wide generation)
            sv.t_pending = now;
            return FX_EUNSUPPORTED;
        }
    }
    if (sv.running) {
        if (N > sv.cap) return FX_EUNSUPPORTED;
        // the workgroups leave 2 x serve_idle_us after their last request: do not post to a generation that may be on its way out
        if (!sv.fresh && std::chrono::duration<double>(std::chrono::steady_clock::now() - sv.t_post).count() * 1e6 > (double)e->serve_idle_us)
            server_stop(e);
        {
            // leaving by themselves (idle / lifetime): all go.  A slot counts as gone when it HAS been seen alive and no longer
            // is -- the workgroups of a fresh generation raise their `alive` words as they start, the late ones microseconds
            // after the first request was answered.  Looked at for the slots this request needs and one rotating slot.
            auto need = [&](int) { return (int)std::min<int64_t>((N + 15) / 16, sv.tiles); };
            auto left = [&](int m, int t) {
                uint8_t& seen = sv.seen[(size_t)m * FX_SERVE_TILES + t];
                if (sv.h_out->alive[m][t]) { seen = 1; return false; }
                return seen != 0;
            };
            for (int m = 0; m < M && sv.running; ++m) {
                for (int t = 0, nt = need(m); t < nt; ++t)
                    if (left(m, t)) { server_stop(e); break; }
                if (sv.running && left(m, (int)(sv.seq % (unsigned)sv.tiles))) server_stop(e);
            }
        }
        // (a generation is replaced well before its workgroups' own lifetime limit)
        if (sv.running && std::chrono::duration<double>(std::chrono::steady_clock::now() - sv.t_start).count() > 4.0) server_stop(e);
    }
    if (!sv.running) {
        // residency pays when calls come densely: start when the same ensemble asks again within the idle window (a caller
        // with milliseconds of host work between its calls would pay a start per call and is better served by launches)
        const auto now = std::chrono::steady_clock::now();
        const bool again = (int)sv.pending.size() == M && std::equal(sv.pending.begin(), sv.pending.end(), models) &&
                           std::chrono::duration<double>(now - sv.t_pending).count() * 1e6 < (double)e->serve_idle_us;
        sv.t_pending = now;
        if (!again) { sv.pending.assign(models, models + M); return FX_EUNSUPPORTED; }
        const int rc = server_start(e, models, M, L, lut);
        if (rc) {
            sv.pending.clear();
            sv.refused.assign(models, models + M); sv.refused_L = L;   // (no resident form, or the start failed: do not try again per call)
            (void)hipGetLastError();
            return FX_EUNSUPPORTED;
        }
        if (N > sv.cap) return FX_EUNSUPPORTED;
    }
    sv.prof_ns[0] = since();
    return FX_OK;
}

// Post a request to the running generation.  `ascii` given: bytes, fence, request word, fence (write-combining stores may pass
// each other otherwise).  `ascii` null: a STREAMED request -- `ready` = no rows yet, then the request word with FX_SERVE_STREAM;
// the caller packs rows into sv.in->bytes and reports them with server_rows_ready.
static void server_post(fx_engine* e, const uint8_t* ascii, int64_t N, int L) {
    auto& sv = e->server;
    if ((++sv.seq & 0x7FFFFFFFull) == 0) ++sv.seq;         // 31-bit tags, never 0; they run on across generations, so a slot's stale answer never matches
    const unsigned seq = (unsigned)(sv.seq & 0x7FFFFFFFull);
    unsigned long long word = ((unsigned long long)seq << 16) | (unsigned long long)N;
    if (ascii && e->serve_tiny && N <= 16 && (size_t)N * L <= FX_SERVE_TINY_BYTES) {      // (one tile: only its workgroup reads the line)
        // a tiny request: its bytes and a second copy of the word go into the request word's own line (FxMailIn::tiny); the slot
        // of tile 0 reads the whole line per poll and needs no second read for the bytes
        word |= FX_SERVE_TINY;
        unsigned char pad[FX_SERVE_TINY_BYTES] = {};
        std::memcpy(pad, ascii, (size_t)N * L);
        std::memcpy(sv.in->tiny, pad, sizeof pad);         // (the whole 48 bytes: full write-combining lines, and no stale bytes behind the request's)
        sv.in->req_tail = word;
    } else if (ascii) {
        std::memcpy(sv.in->bytes, ascii, (size_t)N * L);
    } else {
        sv.in->ready = (unsigned long long)seq << 16; word |= FX_SERVE_STREAM;
    }
    fx_bar_fence();
    // (the slots beyond the fast ones poll the copy: written first -- a slot that sees it early finds the bytes in place all the same)
    if (sv.fast < sv.tiles) sv.in->req_wide = word;
    sv.in->req = word;
    fx_bar_fence();
    sv.t_post = std::chrono::steady_clock::now();
    sv.posted_N = N;
    sv.prof_ns[1] = server_since(e);
}

static void server_rows_ready(fx_engine* e, int64_t rows) {
    auto& sv = e->server;
    fx_bar_fence();                                        // the rows' bytes before the word that announces them
    sv.in->ready = ((sv.seq & 0x7FFFFFFFull) << 16) | (unsigned long long)rows;
    fx_bar_fence();                                        // (and out of the write-combining buffer now)
}

// Collect the posted request's answers.  FX_OK / FX_EBADCHAR: answered.  FX_EUNSUPPORTED: the generation did not answer
// (it was stopped here); the caller launches as usual.
static int server_collect(fx_engine* e, int M, float* out_NM, float* out_mean) {
    auto& sv = e->server;
    auto since = [&]() { return server_since(e); };
    const int64_t N = sv.posted_N;
    const unsigned seq = (unsigned)(sv.seq & 0x7FFFFFFFull);
    const auto t0 = sv.t_post;
    // (the first request of a generation also waits for the launch and the weight fill -- and the very first one of the
    //  process for the runtime to create the high-priority hardware queue and load the kernels: ~0.3 s, once)
    // (later requests: a resident workgroup answers within microseconds and one that left says so through its `alive` word;
    //  the limit only catches a device that has stopped making progress -- or is time-sliced away to another process)
    double limit = sv.fresh ? 3.0 : 0.02;
    for (int m = 0; m < M && limit < 1.0; ++m)
        for (int64_t t = 0; t < std::min<int64_t>((N + 15) / 16, sv.tiles); ++t)
            if (!sv.seen[(size_t)m * FX_SERVE_TILES + t]) { limit = 3.0; break; }   // (a slot that has never answered may still be starting)
    const FxMailOut* h = sv.h_out;
    bool bad = false;
    // Collected MEMBER BY MEMBER into planes: every answer line was just written by the device, i.e. is a cache miss for
    // this core, and a (sequence, member) walk touches M streams at once with nothing requested ahead -- 32 ns per sequence
    // for three members, more than the device needed for a 1000-sequence request (profiles/r4_server_wide_ab_first.log).
    // One sequential stream at a time with the lines eight ahead requested early costs a few ns per answer; by the time
    // member 0 is through, the other members' answers have usually all landed.
    if (sv.planes.size() < (size_t)M * (size_t)N) sv.planes.resize((size_t)M * (size_t)N);
    float* pl = sv.planes.data();
    for (int m = 0; m < M; ++m) {
        const volatile unsigned long long* am = h->ans[m];
        float* pm = pl + (size_t)m * (size_t)N;
        // (lines ahead: 8 while the device is still answering -- a line requested before the device writes it comes back stale
        //  and is missed again -- and 32 for the later members, whose answers have mostly landed by the time their turn comes)
        const int64_t ahead = m == 0 ? 64 : 256;
        if (m > 0) for (int64_t n = 0; n < ahead && n < N; n += 8) __builtin_prefetch(const_cast<const unsigned long long*>(am) + n, 0, 0);
        int vbad = 0;
        for (int64_t n = 0; n < N; ++n) {
            if ((n & 7) == 0) {
                // whole lines whose eight answers have all arrived: vector path (host_collect.cc); the loop below waits for the rest
                const int64_t n2 = fx_collect_lines(am, pm, n, N, seq, ahead, &vbad);
                if (n2 > n) {
                    if (m == 0 && n == 0) sv.prof_ns[2] = since();
                    for (int64_t t = n >> 4; t <= (n2 - 1) >> 4; ++t) sv.seen[(size_t)m * FX_SERVE_TILES + (size_t)(t % sv.tiles)] = 1;
                    n = n2 - 1;
                    continue;
                }
                __builtin_prefetch(const_cast<const unsigned long long*>(am) + n + ahead, 0, 0);
            }
            unsigned spins = 0;
            unsigned long long a;
            while ((((a = am[n]) >> 32) & 0x7FFFFFFFull) != seq) {
                __builtin_ia32_pause();                     // (spin-wait hint: leaves the core's resources to a sibling hyperthread)
                if ((++spins & 1023u) == 0) {
                    const int slot = (int)((n >> 4) % sv.tiles);
                    const bool gone = sv.seen[(size_t)m * FX_SERVE_TILES + slot] && !h->alive[m][slot];
                    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    // (this thread may have been off the core for milliseconds between the read above and this clock: look again
                    //  before giving up on an answer that has arrived meanwhile)
                    if ((gone || waited > limit) && (((a = am[n]) >> 32) & 0x7FFFFFFFull) == seq) break;
                    if (gone || waited > limit) {
                        server_stop(e);                    // fall back to a launch; the next calls start a new generation
                        sv.fallbacks += 1;
                        sv.fb_info = (gone ? 1000000000ll : 2000000000ll) + (int64_t)m * 10000000 + (n % 1000) * 10000 + (int64_t)std::min(waited * 1e6, 9999.0);
                        return FX_EUNSUPPORTED;
                    }
                }
            }
            if (m == 0 && n == 0) sv.prof_ns[2] = since();
            bad = bad || (a >> 63);
            const unsigned bits = (unsigned)a;
            std::memcpy(&pm[n], &bits, 4);
            if ((n & 15) == 0) sv.seen[(size_t)m * FX_SERVE_TILES + (size_t)((n >> 4) % sv.tiles)] = 1;   // (it answered: it is there)
        }
        bad = bad || vbad;
        if (m < 3) sv.prof_ns[5 + m] = since();
    }
    sv.prof_ns[3] = since();
    if (out_NM) {
        for (int m = 0; m < M; ++m) {
            const float* pm = pl + (size_t)m * (size_t)N;
            for (int64_t n = 0; n < N; ++n) out_NM[n * M + m] = pm[n];
        }
    }
    if (out_mean) host_mean_planes(pl, N, N, M, out_mean);
    sv.prof_ns[4] = since();
    sv.fresh = false;
    sv.served += 1;
    e->counters.host_calls += 1; e->counters.zero_copy_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    if (bad) return fx_fail(e, FX_EBADCHAR, "substring not found: character outside the alphabet");
    return FX_OK;
}

// FX_OK: answered.  FX_EUNSUPPORTED: not this time (the caller launches as usual).  Anything else: the call's error.
static int server_call(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L,
                       const uint8_t lut[256], float* out_NM, float* out_mean) {
    const int rc = server_admit(e, models, M, N, L, lut);
    if (rc) return rc;
    server_post(e, ascii, N, L);
    return server_collect(e, M, out_NM, out_mean);
}

// ---- streamed calls (round 4): the request is posted first, the caller packs its strings straight into the mailbox ----------
int fx_score_stream_begin(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], uint8_t** rows_out) {
    if (!e) return FX_EINVAL;
    if (!rows_out) return fx_fail(e, FX_EINVAL, "null buffer");
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 1) return FX_EUNSUPPORTED;
    if (e->server.streaming) return fx_fail(e, FX_ESTATE, "fx_score_stream_begin: a streamed call is already open");
    {
        // only a call that a RUNNING generation of exactly these members can take is streamed; everything else -- starting a
        // generation, the adaptive change of geometry -- is the packed call's business (server_call), whose bookkeeping a
        // refused attempt here must not touch
        const auto& sv = e->server;
        bool same = sv.running && !sv.fresh && N <= sv.cap && (int)sv.models.size() == M && sv.L == L && std::memcmp(sv.lut, lut, 256) == 0;
        for (int m = 0; same && m < M; ++m) same = sv.models[m] == models[m] && sv.versions[m] == models[m]->version;
        if (!same) return FX_EUNSUPPORTED;
    }
    FX_HIP(e, hipSetDevice(e->device));
    rc = server_admit(e, models, M, N, L, lut);
    if (rc) return rc;
    server_post(e, nullptr, N, L);
    e->server.streaming = true;
    e->server.stream_M = M;
    *rows_out = e->server.in->bytes;
    return FX_OK;
}

int fx_score_stream_rows(fx_engine* e, int64_t rows) {
    if (!e) return FX_EINVAL;
    if (!e->server.streaming || rows < 0 || rows > e->server.posted_N) return fx_fail(e, FX_ESTATE, "fx_score_stream_rows: no streamed call open, or more rows than announced");
    server_rows_ready(e, rows);
    return FX_OK;
}

int fx_score_stream_end(fx_engine* e, int ok, float* out_NM, float* out_mean) {
    if (!e) return FX_EINVAL;
    if (!e->server.streaming) return fx_fail(e, FX_ESTATE, "fx_score_stream_end: no streamed call open");
    e->server.streaming = false;
    if (!ok) { server_stop(e); return FX_OK; }             // the caller could not pack its rows: the workgroups abandon the request and leave
    if (!out_NM && !out_mean) { server_stop(e); return fx_fail(e, FX_EINVAL, "null buffer"); }
    server_rows_ready(e, e->server.posted_N);
    const int rc = server_collect(e, e->server.stream_M, out_NM, out_mean);
    if (rc == FX_OK) e->server.streamed += 1;
    return rc;
}

int fx_score(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L,
             const uint8_t lut[256], float* out_NM, float* out_mean) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 0) return fx_fail(e, FX_EINVAL, "negative batch size");
    if (N == 0) return FX_OK;
    if (!ascii || (!out_NM && !out_mean)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    if (e->lp_armed.on) {
        // a pre-launched instance of ANOTHER call leaves now, before this call's buffer management can wait for it on the stream
        const auto& pa = e->lp_armed;
        bool same = (int)pa.models.size() == M && pa.N == N && pa.L == L;
        for (int m = 0; same && m < M; ++m) same = pa.models[m] == models[m];
        if (!same) lp_disarm(e);
    }
    {
        rc = server_call(e, models, M, ascii, N, L, lut, out_NM, out_mean);
        if (rc != FX_EUNSUPPORTED) return rc;
    }
    // characters outside the alphabet are detected on the device (deferred error word)
    const size_t in_bytes = (size_t)N * (size_t)L;
    const size_t nm_bytes = sizeof(float) * (size_t)N * (size_t)M, mean_bytes = sizeof(float) * (size_t)N;
    void *d_in = nullptr, *h_in = nullptr, *h_out = nullptr;
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_pinned(e, 0, in_bytes, &h_in))) return rc;
    // mean only: member-major planes as the intermediate (see fx_score_dev)
    const int64_t stride = (!out_NM && M <= 16) ? planar_stride_for(N) : 0;
    const size_t inter_bytes = stride ? sizeof(float) * (size_t)stride * (size_t)M : nm_bytes;
    // explorer-size mean-only calls that are LAUNCHED (the protein CNN, shapes without a resident form): the member planes go
    // straight to pinned host memory and the mean is taken here -- M x N floats over PCIe instead of N, and no second launch
    // (~6 us of a 50 us call)
    const bool host_mean = stride && M > 1 && N <= e->host_mean_below;
    if ((rc = fx_pinned(e, 1, std::max(nm_bytes, host_mean ? inter_bytes : (size_t)0) + mean_bytes, &h_out))) return rc;
    void* d_out = nullptr;
    if ((rc = fx_scratch(e, 1, inter_bytes + mean_bytes, &d_out))) return rc;
    float* d_NM = (float*)d_out;
    float* d_mean = (float*)((char*)d_out + inter_bytes);
    if (ascii != h_in) std::memcpy(h_in, ascii, in_bytes);   // (fx_staging_input callers marshalled straight into it)
    if ((rc = fx_upload_lut(e, lut))) return rc;
    bool plan_zc = false;
    int plan_pieces = 1;
    plan_host_call(e, models, M, N, L, &plan_zc, &plan_pieces);
    e->counters.host_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    if (in_bytes + nm_bytes + mean_bytes <= (size_t)e->zero_copy_bytes || plan_zc) {
        e->counters.zero_copy_calls += 1;
        // Small call (what Adalead / CMA-ES / DynaPPO issue): zero-copy through the mapped pinned
        // staging buffers -- the kernels read the sequences from, and write the scores to, host memory over
        // PCIe; two memcpy enqueues and their latencies disappear from the call.
        void *dm_in = nullptr, *dm_out = nullptr;
        FX_HIP(e, hipHostGetDevicePointer(&dm_in, h_in, 0));
        FX_HIP(e, hipHostGetDevicePointer(&dm_out, h_out, 0));
        float* m_NM = (float*)dm_out;
        float* m_mean = (float*)((char*)dm_out + nm_bytes);
        if (host_mean) {
            e->call_prof_ns[0] = server_since(e);
            // a pre-launched instance of exactly this call (the caller is back within the idle window): no launch, no weight fill
            bool served = lp_armed_matches(e, models, M, N, L, lut, stride, 1) && lp_serve_armed(e, models, M, ascii, N, L, lut, m_NM, stride, 1, h_out, inter_bytes);
            if (!served) {
                if ((rc = score_dispatch(e, models, M, (const uint8_t*)dm_in, N, L, m_NM, stride))) return rc;
                e->call_prof_ns[1] = server_since(e);
                // answered by the layer-parallel form alone: its NEXT instance is enqueued now, beside this one's work
                const bool lp = e->done_armed && e->lp_launches == 1 && e->dispatch_groups == 1;
                const unsigned seq = lp ? e->done_seq : 0;
                if (lp) lp_arm(e, models, M, N, L, lut, m_NM, stride, 1);
                if ((rc = wait_for_results(e, seq))) return rc;
            }
            e->call_prof_ns[2] = server_since(e);
            if ((rc = check_deferred(e))) return rc;
            host_mean_planes((const float*)h_out, stride, N, M, out_mean);
            e->call_prof_ns[3] = server_since(e);
            return FX_OK;
        } else if (stride) {
            if ((rc = score_then_mean(e, models, M, (const uint8_t*)dm_in, N, L, d_NM, stride, m_mean))) return rc;
            e->done_armed = false;                         // (the mean kernel, or a fused mean, is the last writer)
        } else {
            const bool plain_matrix = out_NM && !out_mean && N <= e->host_mean_below;
            const bool served = plain_matrix && lp_armed_matches(e, models, M, N, L, lut, 0, 2) && lp_serve_armed(e, models, M, ascii, N, L, lut, m_NM, 0, 2, h_out, nm_bytes);
            unsigned seq = 0;
            if (!served) {
                if ((rc = score_dispatch(e, models, M, (const uint8_t*)dm_in, N, L, out_NM ? m_NM : d_NM, stride))) return rc;
                if (out_mean) {
                    if ((rc = fx_launch_ensemble_reduce(e, out_NM ? m_NM : d_NM, N, M, nullptr, m_mean, nullptr))) return rc;
                    e->done_armed = false;
                }
                if (plain_matrix && e->done_armed && e->lp_launches == 1 && e->dispatch_groups == 1) {
                    seq = e->done_seq;
                    lp_arm(e, models, M, N, L, lut, m_NM, 0, 2);
                }
            }
            if (served) e->done_armed = false;
            else if ((rc = wait_for_results(e, seq))) return rc;   // (the last launch's completion flag where it offers one, else the stream)
            if ((rc = check_deferred(e))) return rc;
            if (out_NM) std::memcpy(out_NM, h_out, nm_bytes);
            if (out_mean) std::memcpy(out_mean, (char*)h_out + nm_bytes, mean_bytes);
            return FX_OK;
        }
        if ((rc = wait_for_results(e))) return rc;         // (the mean kernel was the last writer: the stream)
    } else {
        e->counters.bytes_h2d += (int64_t)in_bytes;
        e->counters.bytes_d2h += (int64_t)((out_mean ? mean_bytes : 0) + (out_NM ? nm_bytes : 0));
        FX_HIP(e, hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, e->stream));
        if (stride) rc = score_then_mean(e, models, M, (const uint8_t*)d_in, N, L, d_NM, stride, d_mean);
        else rc = score_dispatch(e, models, M, (const uint8_t*)d_in, N, L, d_NM, stride);
        if (rc) return rc;
        if (out_mean) {
            if (!stride && (rc = fx_launch_ensemble_reduce(e, d_NM, N, M, nullptr, d_mean, nullptr))) return rc;
            FX_HIP(e, hipMemcpyAsync((char*)h_out + nm_bytes, d_mean, mean_bytes, hipMemcpyDeviceToHost, e->stream));
        }
        if (out_NM) FX_HIP(e, hipMemcpyAsync(h_out, d_NM, nm_bytes, hipMemcpyDeviceToHost, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    if ((rc = check_deferred(e))) return rc;
    if (out_NM) std::memcpy(out_NM, h_out, nm_bytes);
    if (out_mean) std::memcpy(out_mean, (char*)h_out + nm_bytes, mean_bytes);
    return FX_OK;
}

// ---- the same call in pieces: the caller fills the pinned staging area chunk by chunk and submits each chunk
// as soon as it is ready, so that marshalling chunk k+1 on the host overlaps transfer + scoring of chunk k.
int fx_score_begin(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256],
                   int want_nm, int want_mean, void** staging) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 1 || !staging || (!want_nm && !want_mean)) return fx_fail(e, FX_EINVAL, "fx_score_begin: bad arguments");
    if (e->chunked.active) return fx_fail(e, FX_ESTATE, "fx_score_begin: a chunked call is already in flight");
    FX_HIP(e, hipSetDevice(e->device));
    const size_t in_bytes = (size_t)N * (size_t)L;
    const size_t nm_bytes = sizeof(float) * (size_t)N * (size_t)M, mean_bytes = sizeof(float) * (size_t)N;
    void *d_in = nullptr, *h_in = nullptr, *h_out = nullptr, *d_out = nullptr;
    if ((rc = fx_scratch(e, 0, in_bytes + 16, &d_in))) return rc;
    if ((rc = fx_pinned(e, 0, in_bytes, &h_in))) return rc;
    if ((rc = fx_pinned(e, 1, nm_bytes + mean_bytes, &h_out))) return rc;
    if ((rc = fx_scratch(e, 1, nm_bytes + mean_bytes, &d_out))) return rc;
    if ((rc = fx_upload_lut(e, lut))) return rc;
    auto& c = e->chunked;
    c.models.assign(models, models + M);
    c.N = N; c.L = L; c.want_nm = want_nm != 0; c.want_mean = want_mean != 0;
    c.h_in = (uint8_t*)h_in; c.d_in = (uint8_t*)d_in;
    c.d_nm = (float*)d_out; c.d_mean = (float*)((char*)d_out + nm_bytes);
    c.h_out = (char*)h_out;
    c.pieces = 0;
    { int unused = 1; plan_host_call(e, models, M, N, L, &c.zero_copy, &unused); }
    c.active = true;
    *staging = h_in;
    return FX_OK;
}

int fx_score_submit(fx_engine* e, int64_t row0, int64_t rows) {
    if (!e) return FX_EINVAL;
    auto& c = e->chunked;
    if (!c.active) return fx_fail(e, FX_ESTATE, "fx_score_submit without fx_score_begin");
    if (row0 < 0 || rows < 0 || row0 + rows > c.N) return fx_fail(e, FX_EINVAL, "fx_score_submit: rows out of range");
    if (rows == 0) return FX_OK;
    FX_HIP(e, hipSetDevice(e->device));
    const int M = (int)c.models.size();
    const size_t nm_bytes = sizeof(float) * (size_t)c.N * (size_t)M;
    int rc;
    // transfers on the copy stream, kernels on the compute stream, one event triple per piece: the upload of piece k + 1
    // (packed by the host while piece k runs) and the download of piece k - 1 overlap piece k's kernels
    if (c.pieces >= fx_engine::MAX_PIECES) return fx_fail(e, FX_EINVAL, "fx_score_submit: more than 32 pieces in one call");
    if (c.pieces == 0) { e->counters.host_calls += 1; e->counters.zero_copy_calls += c.zero_copy ? 1 : 0; }
    e->counters.sequences += rows; e->counters.forwards += rows * M;
    if (!c.zero_copy) {
        e->counters.bytes_h2d += rows * c.L;
        e->counters.bytes_d2h += (int64_t)sizeof(float) * rows * ((c.want_mean ? 1 : 0) + (c.want_nm ? M : 0));
    }
    if (c.zero_copy) {
        // no copy enqueues at all: the piece's kernels read its bytes from the pinned staging area over PCIe (L bytes
        // per sequence against ~1e5 FLOP: the reads hide behind the MFMA work) and the results land in pinned memory
        void *dm_in = nullptr, *dm_out = nullptr;
        FX_HIP(e, hipHostGetDevicePointer(&dm_in, c.h_in, 0));
        FX_HIP(e, hipHostGetDevicePointer(&dm_out, c.h_out, 0));
        float* m_nm = (float*)dm_out + row0 * M;
        float* m_mean = (float*)((char*)dm_out + nm_bytes) + row0;
        float* nm = c.want_nm ? m_nm : c.d_nm + row0 * M;
        if ((rc = score_dispatch(e, c.models.data(), M, (const uint8_t*)dm_in + row0 * c.L, rows, c.L, nm))) return rc;
        if (c.want_mean && (rc = fx_launch_ensemble_reduce(e, nm, rows, M, nullptr, m_mean, nullptr))) return rc;
        const int k = c.pieces;
        FX_HIP(e, hipEventRecord(e->ev_out[k], e->stream));
        c.row0[k] = row0; c.rows[k] = rows;
        ++c.pieces;
        return FX_OK;
    }
#if defined(FX_AB)
    const bool two = e->chunk_overlap != 0;
#else
    const bool two = false;                               // (the two-stream form measured slower: A/B build only)
#endif
    hipStream_t cs = two ? e->copy_stream : e->stream;
    const int k = c.pieces;
    FX_HIP(e, hipMemcpyAsync(c.d_in + row0 * c.L, c.h_in + row0 * c.L, (size_t)rows * c.L, hipMemcpyHostToDevice, cs));
    if (two) {
        FX_HIP(e, hipEventRecord(e->ev_in[k], cs));
        FX_HIP(e, hipStreamWaitEvent(e->stream, e->ev_in[k], 0));
    }
    float* nm = c.d_nm + row0 * M;
    if ((rc = score_dispatch(e, c.models.data(), M, c.d_in + row0 * c.L, rows, c.L, nm))) return rc;
    if (c.want_mean)
        if ((rc = fx_launch_ensemble_reduce(e, nm, rows, M, nullptr, c.d_mean + row0, nullptr))) return rc;
    if (two) {
        FX_HIP(e, hipEventRecord(e->ev_done[k], e->stream));
        FX_HIP(e, hipStreamWaitEvent(cs, e->ev_done[k], 0));
    }
    if (c.want_mean)
        FX_HIP(e, hipMemcpyAsync(c.h_out + nm_bytes + sizeof(float) * row0, c.d_mean + row0, sizeof(float) * rows,
                                 hipMemcpyDeviceToHost, cs));
    if (c.want_nm)
        FX_HIP(e, hipMemcpyAsync(c.h_out + sizeof(float) * row0 * M, nm, sizeof(float) * rows * M, hipMemcpyDeviceToHost, cs));
    FX_HIP(e, hipEventRecord(e->ev_out[k], cs));
    c.row0[k] = row0; c.rows[k] = rows;
    ++c.pieces;
    return FX_OK;
}

int fx_score_finish(fx_engine* e, float* out_NM, float* out_mean) {
    if (!e) return FX_EINVAL;
    auto& c = e->chunked;
    if (!c.active) return fx_fail(e, FX_ESTATE, "fx_score_finish without fx_score_begin");
    c.active = false;
    FX_HIP(e, hipSetDevice(e->device));
    const int M = (int)c.models.size();
    const size_t nm_bytes = sizeof(float) * (size_t)c.N * (size_t)M;
    // piece by piece: the host copies piece k out of the pinned area while the GPU still works on the later ones
    // (a character outside the alphabet in ANY piece fails the call: the results are only trusted after the last check)
    for (int k = 0; k < c.pieces; ++k) {
        FX_HIP(e, hipEventSynchronize(e->ev_out[k]));
        if (c.want_nm && out_NM)
            std::memcpy(out_NM + c.row0[k] * M, c.h_out + sizeof(float) * c.row0[k] * M, sizeof(float) * (size_t)c.rows[k] * M);
        if (c.want_mean && out_mean)
            std::memcpy(out_mean + c.row0[k], c.h_out + nm_bytes + sizeof(float) * c.row0[k], sizeof(float) * (size_t)c.rows[k]);
    }
    FX_HIP(e, hipStreamSynchronize(e->copy_stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

int fx_encode_onehot_dev(fx_engine* e, const uint8_t* d_ascii, int64_t N, int L, const uint8_t lut[256], int A,
                         float* d_one_hot) {
    if (!e || !lut || N < 0 || L < 0 || A < 1) return FX_EINVAL;
    if (N == 0 || L == 0) return FX_OK;
    if (!d_ascii || !d_one_hot) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    int rc = fx_upload_lut(e, lut);
    if (rc) return rc;
    return fx_launch_encode_onehot(e, d_ascii, N, L, A, d_one_hot);
}

int fx_encode_onehot(fx_engine* e, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], int A,
                     float* one_hot) {
    if (!e || !lut || N < 0 || L < 0 || A < 1) return FX_EINVAL;
    if (N == 0 || L == 0) return FX_OK;
    if (!ascii || !one_hot) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const size_t in_bytes = (size_t)N * L, out_bytes = sizeof(float) * (size_t)N * L * A;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, out_bytes, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, ascii, in_bytes, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_encode_onehot(e, (const uint8_t*)d_in, N, L, A, (float*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(one_hot, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

int fx_ensemble_reduce_dev(fx_engine* e, const float* d_scores, int64_t N, int M, const double* weights,
                           float* d_out32, double* d_out64) {
    if (!e || N < 0 || M < 1) return FX_EINVAL;
    if (N == 0) return FX_OK;
    if (!d_scores || (weights ? !d_out64 : !d_out32)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const double* d_w = nullptr;
    if (weights) {
        void* p = nullptr;
        int rc = fx_scratch(e, 3, sizeof(double) * (size_t)M, &p);
        if (rc) return rc;
        FX_HIP(e, hipMemcpyAsync(p, weights, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));       // `weights` is caller memory
        d_w = (const double*)p;
    }
    return fx_launch_ensemble_reduce(e, d_scores, N, M, d_w, d_out32, d_out64);
}

int fx_ensemble_reduce(fx_engine* e, const float* scores, int64_t N, int M, const double* weights, float* out32,
                       double* out64) {
    if (!e || N < 0 || M < 1) return FX_EINVAL;
    if (N == 0) return FX_OK;
    if (!scores || (weights ? !out64 : !out32)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const size_t in_bytes = sizeof(float) * (size_t)N * M;
    const size_t out_bytes = (weights ? sizeof(double) : sizeof(float)) * (size_t)N;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if (in_bytes + out_bytes <= (size_t)e->zero_copy_bytes) {
        FxZeroCopy z;
        const size_t o_w = (in_bytes + 15) / 16 * 16;
        if ((rc = fx_zero_copy_buffers(e, o_w + sizeof(double) * (size_t)M, out_bytes, &z))) return rc;
        std::memcpy(z.h_in, scores, in_bytes);
        if (weights) std::memcpy(z.h_in + o_w, weights, sizeof(double) * (size_t)M);
        if ((rc = fx_launch_ensemble_reduce(e, (const float*)z.d_in, N, M, weights ? (const double*)(z.d_in + o_w) : nullptr,
                                            (float*)z.d_out, (double*)z.d_out))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(weights ? (void*)out64 : (void*)out32, z.h_out, out_bytes);
        return FX_OK;
    }
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, out_bytes, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, scores, in_bytes, hipMemcpyHostToDevice, e->stream));
    rc = fx_ensemble_reduce_dev(e, (const float*)d_in, N, M, weights, (float*)d_out, (double*)d_out);
    if (rc) return rc;
    FX_HIP(e, hipMemcpyAsync(weights ? (void*)out64 : (void*)out32, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

int fx_argmax_decode(fx_engine* e, const double* one_hot, int64_t P, int L, int A, const uint8_t* alphabet,
                     uint8_t* out_chars) {
    if (!e || P < 0 || L < 0 || A < 1) return FX_EINVAL;
    const int64_t rows = P * L;
    if (rows == 0) return FX_OK;
    if (!one_hot || !alphabet || !out_chars) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const size_t in_bytes = sizeof(double) * (size_t)rows * A;
    void *d_in = nullptr, *d_out = nullptr, *d_al = nullptr;
    int rc;
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, (size_t)rows, &d_out))) return rc;
    if ((rc = fx_scratch(e, 3, 256, &d_al))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, one_hot, in_bytes, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_al, alphabet, (size_t)A, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_launch_argmax_decode(e, (const double*)d_in, rows, A, (const uint8_t*)d_al, (uint8_t*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(out_chars, d_out, (size_t)rows, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

// CMA-ES / DynaPPO population step (cmaes.py:61-67 + 83-93, environments/dyna_ppo.py:144-163): decode P
// solutions to sequences (K6) and score them with the ensemble in ONE device round trip -- the decoded
// characters never leave the GPU between the two steps.
int fx_decode_score(fx_engine* e, fx_model* const* models, int M, const double* one_hot, int64_t P, int L, int A,
                    const uint8_t* alphabet, const uint8_t lut[256], uint8_t* out_chars, float* out_NM,
                    float* out_mean) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (P < 0 || A < 1) return fx_fail(e, FX_EINVAL, "bad population shape");
    if (models[0]->shape.A != A) return fx_fail(e, FX_ESHAPE, "alphabet size does not match the model's");
    if (P == 0 || L == 0) return FX_OK;
    if (!one_hot || !alphabet || !out_chars || (!out_NM && !out_mean)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const int64_t rows = P * L;
    const size_t in_bytes = sizeof(double) * (size_t)rows * A;
    const size_t nm_bytes = sizeof(float) * (size_t)P * (size_t)M, mean_bytes = sizeof(float) * (size_t)P;
    void *d_in = nullptr, *d_out = nullptr, *d_txt = nullptr;
    if (in_bytes + 256 + nm_bytes + mean_bytes + (size_t)rows <= (size_t)e->zero_copy_bytes) {
        // a CMA-ES / DyNA-PPO population (15-40 members): the three kernels read and write mapped pinned memory, one wait
        FxZeroCopy z;
        const size_t o_chars = (nm_bytes + mean_bytes + 15) / 16 * 16;
        if ((rc = fx_zero_copy_buffers(e, in_bytes + 256, o_chars + (size_t)rows + 16, &z))) return rc;
        std::memcpy(z.h_in, one_hot, in_bytes);
        std::memcpy(z.h_in + in_bytes, alphabet, (size_t)A);
        if ((rc = fx_upload_lut(e, lut))) return rc;
        float* z_NM = (float*)z.d_out;
        float* z_mean = (float*)(z.d_out + nm_bytes);
        uint8_t* z_chars = (uint8_t*)(z.d_out + o_chars);
        if ((rc = fx_launch_argmax_decode(e, (const double*)z.d_in, rows, A, (const uint8_t*)(z.d_in + in_bytes), z_chars))) return rc;
        if ((rc = score_dispatch(e, models, M, z_chars, P, L, z_NM))) return rc;
        if (out_mean && (rc = fx_launch_ensemble_reduce(e, z_NM, P, M, nullptr, z_mean, nullptr))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        if (out_mean) std::memcpy(out_mean, z.h_out + nm_bytes, mean_bytes);
        if (out_NM) std::memcpy(out_NM, z.h_out, nm_bytes);
        std::memcpy(out_chars, z.h_out + o_chars, (size_t)rows);
        return check_deferred(e);
    }
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, nm_bytes + mean_bytes, &d_out))) return rc;
    if ((rc = fx_scratch(e, 3, 256 + (size_t)rows + 16, &d_txt))) return rc;
    uint8_t* d_al = (uint8_t*)d_txt;
    uint8_t* d_chars = d_al + 256;
    float* d_NM = (float*)d_out;
    float* d_mean = (float*)((char*)d_out + nm_bytes);
    FX_HIP(e, hipMemcpyAsync(d_in, one_hot, in_bytes, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_al, alphabet, (size_t)A, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_argmax_decode(e, (const double*)d_in, rows, A, d_al, d_chars))) return rc;
    if ((rc = score_dispatch(e, models, M, d_chars, P, L, d_NM))) return rc;
    if (out_mean) {
        if ((rc = fx_launch_ensemble_reduce(e, d_NM, P, M, nullptr, d_mean, nullptr))) return rc;
        FX_HIP(e, hipMemcpyAsync(out_mean, d_mean, mean_bytes, hipMemcpyDeviceToHost, e->stream));
    }
    if (out_NM) FX_HIP(e, hipMemcpyAsync(out_NM, d_NM, nm_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipMemcpyAsync(out_chars, d_chars, (size_t)rows, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

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
