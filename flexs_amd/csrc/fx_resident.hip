// C ABI of libflexs_amd.so, part 4 of 6 (fx_internal.h): the resident small-call server (host side of the mailboxes), the
// pre-launched instance of the layer-parallel protein form, streamed calls, and the ways a call waits for its results.
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

// ----------------------------------------------------------------------------------------------------------------
// Resident small-call form: the host side of the mailboxes (score_cnn_quad.hip, SERVER; FxMailIn / FxMailOut).
void server_stop(fx_engine* e) { fx_server_stop(e); }

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
void host_mean_planes(const float* pl, int64_t stride, int64_t N, int M, float* out_mean) {
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

// ---- launched-first host calls (fx_common.h FxRowsReady): one line of device memory the host can store into ---------------------
unsigned* rows_words_ensure(fx_engine* e) {
    if (e->rows_words) return e->rows_words;
    if (e->rows_refused || !e->large_bar) return nullptr;
    e->rows_refused = true;                                // (asked once)
    unsigned* q = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&q), 256, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    {
        std::lock_guard<std::mutex> lock(g_probe_mu);
        bool ok = host_range_is_writable(q, 256);
        if (ok) {
            volatile unsigned* p = q;
            *p = 0x5EB1A5EDu;
            fx_bar_fence();
            unsigned back = 0;
            ok = hipMemcpy(&back, const_cast<const unsigned*>(p), sizeof back, hipMemcpyDeviceToHost) == hipSuccess && back == 0x5EB1A5EDu;
            if (!ok) (void)hipGetLastError();
        }
        if (!ok) { (void)hipFree(q); return nullptr; }
    }
    for (int i = 0; i < 64; ++i) reinterpret_cast<volatile unsigned*>(q)[i] = 0u;
    fx_bar_fence();
    e->rows_words = q;
    e->rows_refused = false;
    return q;
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
void lp_disarm(fx_engine* e) {
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
void lp_arm(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], float* out_dev, int64_t stride, int mode) {
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
    a.out_dev = out_dev; a.scratch2 = e->d_scratch[2]; a.zero_pool = e->d_zero_pool;
    a.t = std::chrono::steady_clock::now();
}

// Is the pre-launched instance the one this call can use?
bool lp_armed_matches(const fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], int64_t stride, int mode, const void* out_dev) {
    const auto& a = e->lp_armed;
    if (a.on && (a.out_dev != out_dev || a.scratch2 != e->d_scratch[2] || a.zero_pool != e->d_zero_pool)) return false;      // (a buffer the armed launch writes was reallocated since)
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
bool lp_serve_armed(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256],
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
int wait_for_results(fx_engine* e, unsigned want_seq) {
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
int fx_wait_small(fx_engine* e) {
    e->done_armed = false;
    return wait_for_results(e);
}

int64_t server_since(const fx_engine* e) {
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
int server_call(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L,
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

}  // extern "C"
