// C ABI of libflexs_amd.so, part 1 of 6 (fx_internal.h): library, engine, options, device / pinned buffers, timers, models.
// include/flexs_amd.h has the contract and the reference file:line each entry point replaces.
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
        // COHERENT (uncached on the device) host memory: the launched-first forms read rows the host stores while the kernel runs and
        // rely on no device cache level holding a line of the staging area (fx_rows_wait); a non-coherent area was measured in round 5,
        // bought nothing (profiles/r5: 255 / 110 / 340 us either way) and is gone
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
int fx_zero_copy_buffers(fx_engine* e, size_t in_bytes, size_t out_bytes, FxZeroCopy* z) {
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

__global__ void k_error_word(const unsigned* err, float* dst) {
    unsigned v = 0;
    for (int i = 0; i < FX_ERR_WORDS; ++i) v |= __hip_atomic_load(err + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *dst = (float)v;
}

int check_deferred(fx_engine* e) {
    // caller has synchronised the stream; the error word lives in mapped pinned host memory,
    // so reading it costs nothing (no extra hipMemcpy on the small-call latency path)
    const unsigned err = fx_err_read(e->h_err);
    if (err) {
        fx_err_clear(e->h_err);
        if (err & FX_ERR_TIMEOUT) {
            // (the barrier counter no longer matches what the host has added up: start over)
            if (e->d_lp_bar) { (void)hipMemset(e->d_lp_bar, 0, FX_LP_BAR_BYTES); for (unsigned& t : e->lp_bar_total) t = 0; }
            if (e->d_zero_pool) (void)hipMemset(e->d_zero_pool, 0, e->zero_pool_bytes);   // (partial maxima / tickets may be left behind)
            return fx_fail(e, FX_ESTATE, "a device-side barrier of the layer-parallel CNN form timed out (workgroups not co-resident?): set the engine option cnn_lp = 0");
        }
        if (err & FX_ERR_STARVED) return fx_fail(e, FX_ESTATE, "rows of a launched-first call never reached the kernels (launch_first = 0 turns the form off)");
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
    fx_err_clear(e->h_err);
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
    if (e->rows_words) (void)hipFree(e->rows_words);
    if (e->relay_flags) (void)hipFree(e->relay_flags);
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

// The deferred error words of this engine, OR-ed, as ONE float at `d_dst` -- enqueued on the engine's stream, i.e. it reports every kernel
// enqueued before it.  flexs_amd/distributed.py puts it into the padding of the block a rank sends into the all-gather, so that "a
// character outside the alphabet on ANY rank" travels with the scores (one collective per call, no flag all-reduce, no host sync).
// Exact: the word is < 8.  Does not clear the words: fx_engine_sync still reports them on this rank.
int fx_engine_error_word_dev(fx_engine* e, float* d_dst) {
    if (!e || !d_dst) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_error_word, dim3(1), dim3(1), 0, e->stream, e->d_err, d_dst);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
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
    if (!std::strcmp(key, "mlp_l1_pos")) return &e->mlp_l1_pos;
    if (!std::strcmp(key, "mlp_l1_pos_tiles")) return &e->mlp_l1_pos_tiles;
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
    if (!std::strcmp(key, "dense_slab_coop")) return &e->dense_slab_coop;
    if (!std::strcmp(key, "cnn_big_units")) return &e->cnn_big_units;
    if (!std::strcmp(key, "poison_outputs")) return &e->poison_outputs;
    if (!std::strcmp(key, "trace")) return &e->trace;
    if (!std::strcmp(key, "ge_bytetab")) return &e->ge_bytetab;
    if (!std::strcmp(key, "wave_prio")) return &e->wave_prio;
    if (!std::strcmp(key, "cnn_quad_tail")) return &e->cnn_quad_tail;
    if (!std::strcmp(key, "cnn_head_slab")) return &e->cnn_head_slab;
    if (!std::strcmp(key, "dense_pipe")) return &e->dense_pipe;
    if (!std::strcmp(key, "fuse_mean")) return &e->fuse_mean;
    if (!std::strcmp(key, "fuse_mean_batch")) return &e->fuse_mean_batch;
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
    if (!std::strcmp(key, "launch_first")) return &e->launch_first;
    if (!std::strcmp(key, "cnn_stage_host")) return &e->cnn_stage_host;
    if (!std::strcmp(key, "launch_relay")) return &e->launch_relay;
    if (!std::strcmp(key, "relay_spread")) return &e->relay_spread;
    if (!std::strcmp(key, "dense_prefetch")) return &e->dense_prefetch;
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
static bool ab_only_value(const fx_engine* e, const int64_t* s, int64_t value) {
#if defined(FX_AB)
    (void)e; (void)s; (void)value;
    return false;
#else
    if (s == &e->dense_pipe || s == &e->fuse_mean || s == &e->fuse_mean_batch || s == &e->chunk_overlap || s == &e->cnn_conv1_mfma || s == &e->mlp_l1_mfma ||
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
    if (e && key && !std::strcmp(key, "launch_first_calls")) { *value = e->launch_first_calls; return FX_OK; }
    if (e && key && !std::strcmp(key, "launch_first_redone")) { *value = e->launch_first_redone; return FX_OK; }
    if (e && key && !std::strcmp(key, "launch_relay_calls")) { *value = e->launch_relay_calls; return FX_OK; }
    if (e && key && !std::strcmp(key, "large_bar")) { *value = (e->large_bar && !e->rows_refused) ? 1 : 0; return FX_OK; }   // (the host can store into device memory: resident forms, launched-first calls)
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

}  // extern "C"
