// C ABI of libflexs_amd.so, part 2 of 6 (fx_internal.h): the scoring entry points -- fx_score and its device / planes / piecewise
// forms -- and the launch planner.  (One-hot encode, ensemble reduction, argmax decode (+ score): fx_codec.hip.)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

#include <sys/mman.h>

#include "fx_common.h"
#include "fx_internal.h"
#include "myers.h"
#include "np_sum.h"
#include <atomic>
#include <mutex>
#include <chrono>

extern "C" {

// ------------------------------------------------------------------ scoring

// (for the launchers: the sequences of the launches enqueued while this lives are read from pinned host memory)
struct HostBytes {
    fx_engine* e;
    explicit HostBytes(fx_engine* e_) : e(e_) { e->ascii_host = true; }
    ~HostBytes() { e->ascii_host = false; }
};

// planar_stride == 0: d_NM is the row-major (N, M) matrix of the ABI; > 0: M member planes that far apart.
int score_dispatch(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                          float* d_NM, int64_t planar_stride) {
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
        if (e->rows_req.on) {
            // launched-first host call: ONE launch of a kernel that waits for its rows, or nothing at all (the caller then packs first)
            if (cnt != M || e->force_generic) return FX_EUNSUPPORTED;
            if (s0.kind == FX_CNN) return fx_launch_score_cnn_mfma(e, models, M, d_ascii, N, d_NM, M, 0);
            return fx_launch_score_dense_mfma(e, models, M, d_ascii, N, d_NM, M, 0);
        }
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
    e->fuse_mean_batch_out = (e->fuse_mean_batch && stride && M > 1 && M <= 16) ? mean_dst : nullptr;
    e->fused_mean_done = false;
    const int rc = score_dispatch(e, models, M, d_ascii, N, L, d_planes, stride);
    e->fuse_mean_out = nullptr;
    e->fuse_mean_batch_out = nullptr;
    if (rc) return rc;
    if (e->fused_mean_done) return FX_OK;
    return fx_launch_ensemble_mean_planar(e, d_planes, N, M, stride, mean_dst);
}

int validate_models(fx_engine* e, fx_model* const* models, int M, int L, const uint8_t* lut) {
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
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    return fx_launch_ensemble_mean_planar(e, d_planes, N, M, stride, d_out_mean);
}

// ... and both halves as one call: the scoring kernel takes the mean itself where a launcher offers it (round 6: the batch form of the
// 4-letter CNN), otherwise the mean kernel follows.  The planes hold the members' scores afterwards either way.
int fx_score_mean_planes_dev(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L,
                             const uint8_t lut[256], float* d_planes, int64_t stride, float* d_out_mean) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (M > 16) return fx_fail(e, FX_EINVAL, "fx_score_mean_planes_dev: at most 16 members");
    if (N < 0 || stride < N || (stride & 3)) return fx_fail(e, FX_EINVAL, "fx_score_mean_planes_dev: stride must be >= N and a multiple of 4");
    if (N == 0) return FX_OK;
    if (!d_ascii || !d_planes || !d_out_mean || (reinterpret_cast<uintptr_t>(d_planes) & 15)) return fx_fail(e, FX_EINVAL, "null or unaligned buffer");
    FX_HIP(e, hipSetDevice(e->device));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    e->counters.device_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    return score_then_mean(e, models, M, d_ascii, N, L, d_planes, stride, d_out_mean);
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
    // an MLP that takes its first layer position-major at batch size wants its bytes in device memory (score_dense_l1.h reads two to four
    // bytes per sequence and barrier); the gather form it would run over host rows is 2-3 x slower than the copy + that path
    for (int m = 0; zc && m < M; ++m)
        if (fx_mlp_l1_pos_applies(e, models[m]->shape, models[m]->layout) && (int64_t)M * ((N + 15) / 16) > (int64_t)e->num_cus * e->mlp_l1_pos_tiles) zc = false;
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

static int relay_prepare(fx_engine* e, int64_t TG);
static int staged_enqueue(fx_engine* e, bool* waits);
static bool ascii_rows_relay_ok(const fx_engine* e) { return e->large_bar && !e->rows_refused; }

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
        HostBytes host_bytes(e);
        if (host_mean) {
            e->call_prof_ns[0] = server_since(e);
            // a pre-launched instance of exactly this call (the caller is back within the idle window): no launch, no weight fill
            bool served = lp_armed_matches(e, models, M, N, L, lut, stride, 1, m_NM) && lp_serve_armed(e, models, M, ascii, N, L, lut, m_NM, stride, 1, h_out, inter_bytes);
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
            const bool served = plain_matrix && lp_armed_matches(e, models, M, N, L, lut, 0, 2, m_NM) && lp_serve_armed(e, models, M, ascii, N, L, lut, m_NM, 0, 2, h_out, nm_bytes);
            unsigned seq = 0;
            e->call_prof_ns[0] = server_since(e);
            if (!served) {
                if ((rc = score_dispatch(e, models, M, (const uint8_t*)dm_in, N, L, out_NM ? m_NM : d_NM, stride))) return rc;
                e->call_prof_ns[1] = server_since(e);
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
            e->call_prof_ns[2] = server_since(e);
            if ((rc = check_deferred(e))) return rc;
            if (out_NM) std::memcpy(out_NM, h_out, nm_bytes);
            if (out_mean) std::memcpy(out_mean, (char*)h_out + nm_bytes, mean_bytes);
            e->call_prof_ns[3] = server_since(e);
            return FX_OK;
        }
        if ((rc = wait_for_results(e))) return rc;         // (the mean kernel was the last writer: the stream)
    } else {
        if (e->launch_relay && M >= 2 && !e->chunked.active && ascii_rows_relay_ok(e)) {
            // An ensemble whose members would each read the bytes over PCIe again: no upload in front of the launch -- member 0's
            // workgroups read the staging area and pass every tile on through device memory (FxRelay), the transfer runs beside
            // the scoring.  The launched-first machinery with every stage published: fx_score_finish does the rest.
            unsigned* w = rows_words_ensure(e);
            const int64_t TG = (N + 15) / 16;
            void* d_relay = nullptr;
            const size_t relay_bytes = (size_t)TG * (size_t)((16 * L + 127) / 128 * 128);
            if (w && relay_prepare(e, TG) == FX_OK && fx_scratch(e, 5, relay_bytes + 16, &d_relay) == FX_OK) {
                auto& c = e->chunked;
                c.models.assign(models, models + M);
                c.N = N; c.L = L; c.want_nm = out_NM != nullptr; c.want_mean = out_mean != nullptr;
                c.h_in = (uint8_t*)h_in; c.d_in = (uint8_t*)d_relay;
                c.d_nm = d_NM; c.d_mean = nullptr; c.h_out = (char*)h_out;
                c.pieces = 0; c.zero_copy = true; c.stride = (!out_NM && M <= 16) ? stride : 0;
                e->rows_base += 4096u;
                c.words = w; c.base = e->rows_base; c.lanes = 1; c.pitch = 16 * L; c.packed_ok = true; c.in_place = false; c.relay = true;
                bool waits = false;
                rc = staged_enqueue(e, &waits);
                if (waits || rc == FX_OK) {
                    // (rc == FX_OK without `waits`: a launcher enqueued a kernel that did not take the rows request -- none does today.
                    // Its results are not the relay's: fx_score_finish redoes the launch, as fx_score_begin_staged has it do)
                    c.redo = rc != 0 || !waits;
                    c.staged = true; c.active = true;
                    e->launch_relay_calls += 1;
                    return fx_score_finish(e, out_NM, out_mean);
                }
                if (rc != FX_EUNSUPPORTED) return rc;       // (else nothing was enqueued: the upload below)
            }
        }
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

// "Launch first, pack behind" (round 5).  The call's kernels are enqueued HERE, before a single string has been packed: they read
// the pinned staging area directly (the zero-copy plan) and every wave waits, tile by tile, for the packing lanes to publish the
// stage its tile belongs to (FxRowsReady).  The caller then packs the stages in order -- `_strpack.pack_staged` -- storing
// `base + stages done` into words[lane] after each, and calls fx_score_finish.  The launch latency, the weight fill and most of
// the kernel's run now lie BESIDE the packing instead of behind it (profiles/r5_e2e_breakdown.log: 37 us of packing and 90 us of
// launch + kernel + wait back to back for the 1e5-sequence MLP call).
// The staging area of such a call is TILE-PITCHED: the 16 rows of tile t start at t * pitch, pitch = 16 L rounded up to whole
// 128-byte lines, so that no cache line holds rows of two tiles -- see FxRowsReady.
// FX_EUNSUPPORTED (nothing enqueued, no call in flight): the plan is not zero-copy, the shape's kernel cannot wait for rows, the
// host cannot store into device memory, or there are too few tiles per SIMD to order -- the caller packs first, as before.
// a flag per tile (zero when allocated; a call's value never repeats) and the call's value
static int relay_prepare(fx_engine* e, int64_t TG) {
    if ((size_t)TG > e->relay_flag_words) {
        if (e->relay_flags) { FX_HIP(e, hipStreamSynchronize(e->stream)); (void)hipFree(e->relay_flags); e->relay_flags = nullptr; e->relay_flag_words = 0; }
        const size_t words = (size_t)TG + (size_t)TG / 4 + 1024;
        if (hipMalloc(reinterpret_cast<void**>(&e->relay_flags), words * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return FX_EUNSUPPORTED; }
        FX_HIP(e, hipMemset(e->relay_flags, 0, words * sizeof(unsigned)));
        e->relay_flag_words = words;
    }
    if (++e->relay_seq == 0) {
        // (the value wrapped: flags of 2^32 calls ago could match -- clear them)
        FX_HIP(e, hipStreamSynchronize(e->stream));
        FX_HIP(e, hipMemset(e->relay_flags, 0, e->relay_flag_words * sizeof(unsigned)));
        ++e->relay_seq;
    }
    return FX_OK;
}

static int staged_enqueue(fx_engine* e, bool* waits) {
    auto& c = e->chunked;
    const int M = (int)c.models.size();
    const size_t nm_bytes = c.want_nm ? sizeof(float) * (size_t)c.N * (size_t)M : 0;   // (the results area: the matrix if wanted, then the mean)
    void *dm_in = nullptr, *dm_out = nullptr;
    FX_HIP(e, hipHostGetDevicePointer(&dm_in, c.h_in, 0));
    FX_HIP(e, hipHostGetDevicePointer(&dm_out, c.h_out, 0));
    float* m_NM = (float*)dm_out;
    float* m_mean = (float*)((char*)dm_out + nm_bytes);
    // stages: as many as the SHORTEST per-SIMD share of a workgroup has tiles, so that every share holds a tile of every stage
    // (member m's tiles are cut over ~G / M workgroups, a workgroup's over its four SIMDs); the launcher has the last word
    const int64_t TG = (c.N + 15) / 16, U = TG * M;
    int64_t G = e->grid_blocks > 0 ? e->grid_blocks : e->num_cus;
    if (G > U) G = U;
    const int64_t nb = M > 1 ? (G + M - 1) / M : G;      // workgroups of one member, at most
    e->rows_min_share = TG / (nb > 0 ? nb : 1) / 4;
    e->rows_req.on = true; e->rows_req.used = false; e->rows_req.relay_used = false;
    e->rows_req.r = FxRowsReady{c.words, c.base, c.lanes, 0, c.pitch};
    e->rows_req.relay = c.relay ? FxRelay{c.d_in, e->relay_flags, e->relay_seq, (16 * c.L + 127) / 128 * 128, (int)e->relay_spread} : FxRelay{nullptr, nullptr, 0, 0, 0};
    HostBytes host_bytes(e);
    int rc;
    if (c.stride) rc = score_then_mean(e, c.models.data(), M, (const uint8_t*)dm_in, c.N, c.L, c.d_nm, c.stride, m_mean);
    else {
        rc = score_dispatch(e, c.models.data(), M, (const uint8_t*)dm_in, c.N, c.L, c.want_nm ? m_NM : c.d_nm, 0);
        if (!rc && c.want_mean) rc = fx_launch_ensemble_reduce(e, c.want_nm ? m_NM : c.d_nm, c.N, M, nullptr, m_mean, nullptr);
    }
    *waits = e->rows_req.used;
    e->rows_req.on = false;
    e->rows_req.relay = FxRelay{nullptr, nullptr, 0, 0, 0};
    c.Q = e->rows_req.r.Q;
    e->done_armed = false;                                 // (finish waits on the stream: the mean kernel may be the last writer)
    return rc;
}

int fx_score_begin_staged(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256],
                          int want_nm, int want_mean, int lanes, void** staging, void** words, unsigned* base, int* stages, int* tile_pitch,
                          void* results, int64_t results_bytes) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (N < 1 || !staging || !words || !base || !stages || !tile_pitch || (!want_nm && !want_mean) || lanes < 1 || lanes > 16)
        return fx_fail(e, FX_EINVAL, "fx_score_begin_staged: bad arguments");
    if (e->chunked.active) return fx_fail(e, FX_ESTATE, "fx_score_begin_staged: a chunked call is already in flight");
    if (!e->launch_first) return FX_EUNSUPPORTED;
    FX_HIP(e, hipSetDevice(e->device));
    bool zc = false;
    { int unused = 1; plan_host_call(e, models, M, N, L, &zc, &unused); }
    // every member reading the rows over PCIe again does not hide behind the kernels (the "copy" plan): member 0's workgroups
    // relay them through device memory (FxRelay) where the ensemble's kernel can, else the caller packs, uploads, launches
    const bool relay = !zc;
    if (relay && (!e->launch_relay || M < 2)) return FX_EUNSUPPORTED;
    unsigned* w = rows_words_ensure(e);
    if (!w) return FX_EUNSUPPORTED;
    const int64_t TG = (N + 15) / 16;
    const int pitch = (16 * L + 127) / 128 * 128;
    const size_t in_bytes = (size_t)TG * (size_t)pitch;
    void *d_in = nullptr, *h_in = nullptr, *h_out = nullptr, *d_out = nullptr;
    if (relay) {
        if ((rc = relay_prepare(e, TG))) return rc;
        if ((rc = fx_scratch(e, 0, in_bytes + 16, &d_in))) return rc;
    }
    const size_t nm_bytes = sizeof(float) * (size_t)N * (size_t)M, mean_bytes = sizeof(float) * (size_t)N;
    const int64_t stride = (!want_nm && M <= 16) ? planar_stride_for(N) : 0;
    const size_t inter_bytes = stride ? sizeof(float) * (size_t)stride * (size_t)M : nm_bytes;
    if ((rc = fx_pinned(e, 0, in_bytes, &h_in))) return rc;     // (no device copy of the input: the kernels read this)
    if (results) {
        // results in place: the kernels write into the caller's own pinned buffer (fx_result_alloc) and fx_score_finish copies nothing
        void* probe = nullptr;
        if (results_bytes < (int64_t)((want_nm ? nm_bytes : 0) + (want_mean ? mean_bytes : 0)))
            return fx_fail(e, FX_EINVAL, "fx_score_begin_staged: results buffer too small");
        if (hipHostGetDevicePointer(&probe, results, 0) != hipSuccess) { (void)hipGetLastError(); return fx_fail(e, FX_EINVAL, "fx_score_begin_staged: results is not memory of fx_result_alloc"); }
        h_out = results;
    } else if ((rc = fx_pinned(e, 1, nm_bytes + mean_bytes, &h_out))) return rc;
    if ((rc = fx_scratch(e, 1, inter_bytes + mean_bytes, &d_out))) return rc;
    if ((rc = fx_upload_lut(e, lut))) return rc;
    auto& c = e->chunked;
    c.models.assign(models, models + M);
    c.N = N; c.L = L; c.want_nm = want_nm != 0; c.want_mean = want_mean != 0;
    c.h_in = (uint8_t*)h_in; c.d_in = (uint8_t*)d_in;
    c.d_nm = (float*)d_out; c.d_mean = nullptr;
    c.h_out = (char*)h_out;
    c.pieces = 0; c.zero_copy = true; c.stride = stride;
    e->rows_base += 4096u;
    c.words = w; c.base = e->rows_base; c.lanes = lanes; c.pitch = pitch; c.packed_ok = true; c.in_place = results != nullptr;
    c.relay = relay;
    bool waits = false;
    rc = staged_enqueue(e, &waits);
    if (rc && !waits) return rc;                           // (FX_EUNSUPPORTED: nothing was enqueued)
    c.redo = rc != 0 || !waits;                            // (an enqueue failed half-way: let the queue drain, then try again)
    c.staged = true;
    c.active = true;
    e->counters.host_calls += 1; e->counters.zero_copy_calls += 1; e->counters.sequences += N; e->counters.forwards += N * M;
    e->launch_first_calls += 1;
    if (relay) e->launch_relay_calls += 1;
    *staging = h_in; *words = w; *base = c.base; *stages = c.Q; *tile_pitch = pitch;
    return FX_OK;
}

// Host memory for results that are handed out in place (fx_score_begin_staged's `results`): the caller owns it -- typically a pool
// behind the arrays that wrap it -- and gives it back with fx_result_free when nothing refers to it any more.
// ORDINARY anonymous memory (mmap), then registered with the device (hipHostRegister: pinned + GPU-mapped, the kernels store into it
// over PCIe as they do into hipHostMalloc memory).  Round 5 used hipHostMalloc, whose mapping a fork()ed child does NOT inherit: a child
// that read a result array it inherited would fault, which kept the in-place form opt-in.  An anonymous mapping is inherited
// (copy-on-write) like any NumPy array; the registration stays with the parent.
int fx_result_alloc(fx_engine* e, int64_t bytes, void** host) {
    if (!e || bytes < 1 || !host) return FX_EINVAL;
    FX_HIP(e, hipSetDevice(e->device));
    // whole 2 MiB pages where the kernel grants them (transparent huge pages): the device then translates one page per 2 MiB of scores
    // instead of 512 -- results of 1e5 sequences x 3 members are 1.6 MB
    constexpr size_t HUGE = (size_t)2 << 20;
    const size_t len = (size_t)bytes >= HUGE / 2 ? (((size_t)bytes + HUGE - 1) & ~(HUGE - 1)) : (((size_t)bytes + 4095) & ~(size_t)4095);
    void* p = MAP_FAILED;
    if (len >= HUGE) {
        // an aligned range out of a larger reservation (mmap gives no alignment), the slack returned at once
        char* raw = (char*)mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != MAP_FAILED) {
            char* al = (char*)(((uintptr_t)raw + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
            if (al > raw) munmap(raw, (size_t)(al - raw));
            if (al + len < raw + len + HUGE) munmap(al + len, (size_t)(raw + len + HUGE - (al + len)));
            p = al;
            (void)madvise(p, len, MADV_HUGEPAGE);
        }
    } else {
        p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    }
    if (p == MAP_FAILED) return fx_fail(e, FX_ENOMEM, "mmap of a result buffer failed");
    std::memset(p, 0, len);                                // (touch every page before it is pinned)
    void* dev = nullptr;
    if (hipHostRegister(p, len, hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostUnregister(p); (void)hipGetLastError();
        munmap(p, len);
        return fx_fail(e, FX_ENOMEM, "hipHostRegister of a result buffer failed");
    }
    if (dev != p) {                                        // (the kernels are handed the host address: unified addressing is assumed)
        (void)hipHostUnregister(p);
        munmap(p, len);
        return fx_fail(e, FX_EUNSUPPORTED, "registered host memory is not mapped at its host address on this device");
    }
    e->result_bufs[p] = len;
    *host = p;
    return FX_OK;
}
int fx_result_free(fx_engine* e, void* host) {
    if (!e || !host) return FX_EINVAL;
    auto it = e->result_bufs.find(host);
    if (it == e->result_bufs.end()) return fx_fail(e, FX_EINVAL, "fx_result_free: not a buffer of fx_result_alloc");
    FX_HIP(e, hipSetDevice(e->device));
    FX_HIP(e, hipStreamSynchronize(e->stream));           // (nothing in flight writes into it)
    FX_HIP(e, hipHostUnregister(host));
    munmap(host, it->second);
    e->result_bufs.erase(it);
    return FX_OK;
}

// The caller of a launched-first call could not pack every row (a ragged batch, a non-str item): fx_score_finish then only waits for
// the kernels and drops whatever they raised over the rows that never came.
int fx_score_abandon(fx_engine* e) {
    if (!e) return FX_EINVAL;
    if (!e->chunked.active) return fx_fail(e, FX_ESTATE, "fx_score_abandon without a call in flight");
    e->chunked.packed_ok = false;
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
        HostBytes host_bytes(e);
        if ((rc = score_dispatch(e, c.models.data(), M, (const uint8_t*)dm_in + row0 * c.L, rows, c.L, nm))) return rc;
        if (c.want_mean && (rc = fx_launch_ensemble_reduce(e, nm, rows, M, nullptr, m_mean, nullptr))) return rc;
        const int k = c.pieces;
        // "piece k is done": a word written by the command processor behind the piece's launches and polled in pinned host
        // memory (as wait_for_results does) -- an event record + hipEventSynchronize cost ~30 us per piece, which was more
        // than the overlap of packing with scoring bought (profiles/r5_e2e_breakdown.log)
        c.flag[k] = 0;
        if (e->done_flag && e->d_done) {
            const unsigned v = ++e->done_value ? e->done_value : ++e->done_value;
            if (hipStreamWriteValue32(e->stream, e->d_done + 8, v, 0) == hipSuccess) c.flag[k] = v;
            else (void)hipGetLastError();
        }
        if (!c.flag[k]) FX_HIP(e, hipEventRecord(e->ev_out[k], e->stream));
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
    c.flag[k] = 0;
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
    if (c.staged) {
        c.staged = false;
        // every stage counts as published now, whatever became of the packing (a caller that failed half-way must not leave the
        // kernels waiting; they then run on whatever the staging area holds and the results are dropped by the caller)
        for (int l = 0; l < c.lanes; ++l) reinterpret_cast<volatile unsigned*>(c.words)[l] = c.base + (unsigned)c.Q;
        fx_bar_fence();
        int rc = wait_for_results(e);
        if (rc) return rc;
        const unsigned err = fx_err_read(e->h_err);
        if (err || c.redo) {
            // ANY error (rows that never came were scored as they lay, so a "bad character" beside "starved" says nothing about the
            // caller's strings), or a launch that did not wait.  When the caller did pack everything, the same launch once more -- every
            // stage is published, nothing waits -- gives the answer and the error the reference gives; when it did not, there is
            // nothing to answer
            fx_err_clear(e->h_err);
            e->launch_first_redone += 1;
            if (!c.packed_ok) return FX_OK;                // (the caller raises its own packing error)
            bool waits = false;
            // a relay's flags of the FIRST attempt stand (its copiers gave up waiting and passed the rows on as they lay): the second
            // attempt needs a value of its own, or its readers would take those tiles for delivered (found by the stopped-process
            // test of round 6: "substring not found" for a batch of valid strings)
            if (c.relay && (rc = relay_prepare(e, (c.N + 15) / 16))) return rc;
            if ((rc = staged_enqueue(e, &waits))) return rc == FX_EUNSUPPORTED ? fx_fail(e, FX_ESTATE, "launched-first call: the second attempt found no kernel") : rc;
            if ((rc = wait_for_results(e))) return rc;
            if (c.relay && (fx_err_read(e->h_err) & FX_ERR_STARVED)) {
                // a relay's readers starved again although every row was there: member 0's workgroups did not get onto the device
                // beside them (another process holding CUs).  Third attempt without the relay: every member reads the host rows
                // itself -- slow, but no workgroup waits for another
                fx_err_clear(e->h_err);
                c.relay = false;
                if ((rc = staged_enqueue(e, &waits))) return rc == FX_EUNSUPPORTED ? fx_fail(e, FX_ESTATE, "launched-first call: the third attempt found no kernel") : rc;
                if ((rc = wait_for_results(e))) return rc;
            }
        }
        if ((rc = check_deferred(e))) return rc;
        if (c.in_place) return FX_OK;                      // (the caller's buffer holds the matrix, then the mean)
        const size_t nm_res = c.want_nm ? nm_bytes : 0;
        if (c.want_nm && out_NM) std::memcpy(out_NM, c.h_out, nm_bytes);
        if (c.want_mean && out_mean) std::memcpy(out_mean, c.h_out + nm_res, sizeof(float) * (size_t)c.N);
        return FX_OK;
    }
    // piece by piece: the host copies piece k out of the pinned area while the GPU still works on the later ones
    // (a character outside the alphabet in ANY piece fails the call: the results are only trusted after the last check)
    bool flags_only = c.pieces > 0;
    for (int k = 0; k < c.pieces; ++k) {
        if (c.flag[k]) {
            // (the stream is in order: the word only moves forward, piece k is done when it has reached piece k's value)
            const volatile unsigned* w = e->h_done + 8;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; (int)(*w - c.flag[k]) < 0; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 4095u) == 4095u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
                    FX_HIP(e, hipStreamSynchronize(e->stream));
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            flags_only = false;
            FX_HIP(e, hipEventSynchronize(e->ev_out[k]));
        }
        if (c.want_nm && out_NM)
            std::memcpy(out_NM + c.row0[k] * M, c.h_out + sizeof(float) * c.row0[k] * M, sizeof(float) * (size_t)c.rows[k] * M);
        if (c.want_mean && out_mean)
            std::memcpy(out_mean + c.row0[k], c.h_out + nm_bytes + sizeof(float) * c.row0[k], sizeof(float) * (size_t)c.rows[k]);
    }
    if (!flags_only) {
        FX_HIP(e, hipStreamSynchronize(e->copy_stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));
    }
    e->done_armed = false;
    return check_deferred(e);
}

}  // extern "C"
