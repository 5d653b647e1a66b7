// SIMT emulation of the training step's DEVICE branches (csrc/train_core.h under FXT_EMUL) on the CPU: one host thread per GPU
// thread of a workgroup.
//   * v_mfma_f32_16x16x4_f32 is a rendezvous of the 64 threads of a wave: every lane deposits its A / B element, then computes its
//     four outputs D[4 (l >> 4) + r][l & 15] += sum_k A[row][k] B[k][col] as an fmaf chain over k = 0 .. 3 -- the order the host build
//     of the same source uses, so the two builds must agree BIT FOR BIT;
//   * __syncthreads / s_barrier are pthread barriers over the workgroup, the fences nothing, LDS a heap array of EXACTLY the bytes
//     the host code would ask for, address-space-qualified pointers plain ones.
// What this reaches that the host build cannot: the tile-to-wave dealing of fxt_gemm, its three k-step walks and their overhang
// masks, fxt_gemm_staged's accumulators kept across the staging barriers -- and the barriers themselves: built with
// -fsanitize=thread, a missing barrier between a phase that writes the workspace and one that reads it is a reported data race;
// built with -fsanitize=address, an index past the workspace / the staging buffer is a report.  A lane that leaves a wave's
// uniform control flow before an MFMA deadlocks the rendezvous (the test's timeout).
// Run: simt_train  (prints "simt_train: ok"); tests/test_sanitizers.py builds and runs it.
#include <pthread.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "simt_common.h"

#define FXT_EMUL 1
typedef float emul_f4 __attribute__((ext_vector_type(4)));
struct WaveBox { float a[64], b[64]; pthread_barrier_t bar; };
static thread_local int t_tid = 0;
static WaveBox* g_waves = nullptr;
static pthread_barrier_t g_wg_bar;
static inline void emul_wg_barrier() { pthread_barrier_wait(&g_wg_bar); }
static inline emul_f4 emul_mfma(float a, float b, emul_f4 acc, int, int, int) {
    WaveBox& w = g_waves[t_tid >> 6];
    const int lane = t_tid & 63;
    w.a[lane] = a; w.b[lane] = b;
    pthread_barrier_wait(&w.bar);
    const int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float c = acc[r];
        for (int k = 0; k < 4; ++k) c = std::fmaf(w.a[row + 16 * k], w.b[col + 16 * k], c);
        acc[r] = c;
    }
    pthread_barrier_wait(&w.bar);                          // (the boxes are free for the next instruction)
    return acc;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emul_mfma
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() emul_wg_barrier()
#define __syncthreads() emul_wg_barrier()
#define __hip_atomic_load(p, order, scope) (*(p))
#define __HIP_MEMORY_SCOPE_AGENT 0
static inline unsigned long long wall_clock64() { return 0; }

namespace fxt_emu {
#include "../../flexs_amd/csrc/train_core.h"
}
using namespace fxt_emu;

// One slice = one workgroup of `nthr` threads.  mode 0 / 1 / 2 = train_core.h MODE (plain, rotated rows, + staged conv kernels).
static void run_slice(const FxtJob& j, int mode, int nthr, int slice, const SimtProblem& p, float* lds) {
    const int nw = nthr / 64;
    std::vector<WaveBox> waves((size_t)nw);
    for (auto& w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
    pthread_barrier_init(&g_wg_bar, nullptr, (unsigned)nthr);
    g_waves = waves.data();
    std::vector<std::thread> th;
    th.reserve((size_t)nthr);
    for (int tid = 0; tid < nthr; ++tid)
        th.emplace_back([&, tid]() {
            t_tid = tid;
            const FxtWg wg{tid, nthr};
            const uint8_t* a = p.ascii.data(); const uint8_t* l = p.lut.data(); const float* y = p.labels.data();
            if (mode == 4) fxt_forward_backward<3, 1, FxtDims<0, 20, 32, 100, 5, 1>, 3>(j, wg, 0, slice, a, l, y, lds, (const float*)j.w, (float*)nullptr);
            else if (mode == 3) fxt_forward_backward<3, 1, FxtDimsAny, 3>(j, wg, 0, slice, a, l, y, lds, (const float*)j.w, (float*)nullptr);
            else if (mode == 2) fxt_forward_backward<3, 1, FxtDimsAny, 2>(j, wg, 0, slice, a, l, y, lds, (const float*)j.w, (float*)nullptr);
            else if (mode == 1) fxt_forward_backward<3, 1, FxtDimsAny, 1>(j, wg, 0, slice, a, l, y, lds, (const float*)j.w, (float*)nullptr);
            else fxt_forward_backward<3, 1, FxtDimsAny, 0>(j, wg, 0, slice, a, l, y, lds, (const float*)j.w, (float*)nullptr);
        });
    for (auto& t : th) t.join();
    for (auto& w : waves) pthread_barrier_destroy(&w.bar);
    pthread_barrier_destroy(&g_wg_bar);
}

// One step through the emulated device branches; the updated weights (and the slices' gradient partials).
static std::vector<float> emul_step(const SimtProblem& p, int mode, int nthr, int stage_taps, std::vector<float>* partials) {
    FxtJob j{};
    j.net = fxt_net(p.kind, p.L, p.A, p.kind == 0 ? p.F : 0, p.H, p.kind == 0 ? p.K : 0);
    j.net.ldx = j.net.F;
    j.batch = p.rows; j.steps_per_epoch = 1; j.total_steps = 1; j.n = p.rows; j.R = p.R; j.S = (p.rows + p.R - 1) / p.R;
    std::vector<float> w = p.w, m((size_t)j.net.P, 0.f), v((size_t)j.net.P, 0.f), part((size_t)j.S * (j.net.P + 1), 0.f);
    const float lr = 1e-3f;
    float loss = 0.f;
    j.w = w.data(); j.adam_m = m.data(); j.adam_v = v.data(); j.partial = part.data(); j.order = p.order.data();
    j.keep = p.kind == 0 ? p.keep.data() : nullptr; j.lr_t = &lr; j.step_loss = &loss;
    j.ws_in_lds = 1;
    j.ws_slice = fxt_ws(j.net, p.R).total;
    if (mode == 2) {
        j.ws_slice = fxt_ws(j.net, p.R, true).total + stage_taps * j.net.F * fxt_ld_w(j.net.F);
        j.split_off = stage_taps;
    }
    if (mode >= 3) {                                       // (a tap = 32 rotated rows of 32 floats; mode 4 = the canonical protein instantiation of mode 3)
        j.ws_slice = fxt_ws(j.net, p.R, true).total + stage_taps * 1024;
        j.split_off = stage_taps;
    }
    for (int s = 0; s < j.S; ++s) {
        std::vector<float> lds((size_t)j.ws_slice, 0.f);    // exact size, fresh per workgroup
        run_slice(j, mode, nthr, s, p, lds.data());
    }
    if (partials) *partials = part;
    t_tid = 0;
    fxt_step_loss(j, 0);
    for (int i = 0; i < j.net.P; ++i) fxt_adam(j, 0, i);
    return w;
}

// The CANONICAL instantiations (compile-time dimensions, workspace AND the padded weight image in LDS: what k_train_fb runs for the
// BASELINE surrogates).  canon: train.hip's case numbers.  The image is laid out as the kernel's staging copy lays it out.
static std::vector<float> emul_step_canon(const SimtProblem& p, int canon, int nthr, std::vector<float>* partials) {
    FxtJob j{};
    j.net = fxt_net(p.kind, p.L, p.A, p.kind == 0 ? p.F : 0, p.H, p.kind == 0 ? p.K : 0);       // (padded rows: fxt_ld_x)
    j.batch = p.rows; j.steps_per_epoch = 1; j.total_steps = 1; j.n = p.rows; j.R = p.R; j.S = (p.rows + p.R - 1) / p.R;
    std::vector<float> w = p.w, m((size_t)j.net.P, 0.f), v((size_t)j.net.P, 0.f), part((size_t)j.S * (j.net.P + 1), 0.f);
    const float lr = 1e-3f;
    float loss = 0.f;
    j.w = w.data(); j.adam_m = m.data(); j.adam_v = v.data(); j.partial = part.data(); j.order = p.order.data();
    j.keep = p.kind == 0 ? p.keep.data() : nullptr; j.lr_t = &lr; j.step_loss = &loss;
    j.ws_in_lds = 1; j.w_in_lds = 1; j.canon = canon;
    j.ws_slice = fxt_ws(j.net, p.R).total;
    const FxtLay lay = fxt_lay(j.net, true);
    for (int s = 0; s < j.S; ++s) {
        std::vector<float> lds((size_t)j.ws_slice + (size_t)lay.total, 0.f);
        float* wl = lds.data() + j.ws_slice;
        for (int i = 0; i < j.net.P; ++i) wl[fxt_image_off(j.net, lay, i)] = w[(size_t)i];
        const int nw = nthr / 64;
        std::vector<WaveBox> waves((size_t)nw);
        for (auto& wv : waves) pthread_barrier_init(&wv.bar, nullptr, 64);
        pthread_barrier_init(&g_wg_bar, nullptr, (unsigned)nthr);
        g_waves = waves.data();
        std::vector<std::thread> th;
        for (int tid = 0; tid < nthr; ++tid)
            th.emplace_back([&, tid]() {
                t_tid = tid;
                const FxtWg wg{tid, nthr};
                const uint8_t* a = p.ascii.data(); const uint8_t* l = p.lut.data(); const float* y = p.labels.data();
                float* ws = lds.data(); const float* W = wl;
                switch (canon) {
                    case 1: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8>>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr); break;
                    case 2: fxt_forward_backward<3, 3, FxtDims<1, 4, 0, 100, 0, 8>>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr); break;
                    case 3: fxt_forward_backward<3, 3, FxtDims<2, 20, 0, 100, 0, 8>>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr); break;
                    case 4: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8, 8>>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr); break;
                    case 6: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 4, 14>>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr); break;
                    default: fxt_forward_backward<3, 3>(j, wg, 0, s, a, l, y, ws, W, (float*)nullptr);
                }
            });
        for (auto& t : th) t.join();
        for (auto& wv : waves) pthread_barrier_destroy(&wv.bar);
        pthread_barrier_destroy(&g_wg_bar);
    }
    if (partials) *partials = part;
    t_tid = 0;
    fxt_step_loss(j, 0);
    for (int i = 0; i < j.net.P; ++i) fxt_adam(j, 0, i);
    return w;
}

static int same(const char* what, const std::vector<float>& a, const std::vector<float>& b, const SimtProblem& p, int mode, int nthr, int taps) {
    if (a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0) return 0;
    size_t first = 0;
    while (first < a.size() && first < b.size() && std::memcmp(&a[first], &b[first], 4) == 0) ++first;
    std::printf("%s differ: kind %d L %d A %d F %d K %d rows %d R %d, mode %d, %d threads, %d taps; first at %zu of %zu\n", what, p.kind, p.L, p.A,
                p.F, p.K, p.rows, p.R, mode, nthr, taps, first, a.size());
    return 1;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !std::strcmp(argv[1], "quick");      // (the sanitizer builds: fewer shapes)
    int bad = 0;
    struct Case { int kind, L, A, F, H, K, rows, R, nthr; };
    std::vector<Case> cases = {
        {0, 21, 20, 32, 20, 5, 3, 1, 256},       // protein alphabet, one row per slice: 2 x 2 conv tiles on 4 waves
        {0, 41, 20, 32, 100, 5, 2, 1, 256},      // 3 x 2 conv tiles: the staged form's two tiles per wave
        {0, 12, 4, 32, 100, 5, 10, 5, 256},      // several rows per slice, 4 letters (conv3 has 3 taps), a ragged last slice
        {1, 9, 4, 0, 20, 0, 19, 8, 128},         // MLP: dense layers only, one-hot weight gradient
        {2, 14, 20, 0, 24, 0, 7, 4, 64},         // GlobalEpistasis on one wave
        {0, 68, 4, 32, 20, 5, 1, 1, 512},        // 64 positions: the rotated modes' max-pool shared by 32 threads per channel; 4 x 2 conv tiles on 8 waves
        {0, 37, 20, 32, 100, 5, 1, 1, 1024},      // sixteen waves (the device's workgroup): the F = 32 form's weight gradient deals one job per wave; 19 taps of conv3
    };
    if (!quick) {
        cases.push_back({0, 70, 20, 32, 100, 5, 1, 1, 512});     // 5 x 2 conv tiles on 8 waves
        cases.push_back({0, 23, 20, 64, 9, 4, 2, 1, 512});       // 64 filters (two groups of eight k-steps per tap), even kernel
        cases.push_back({0, 9, 4, 8, 16, 3, 37, 16, 1024});      // the full 16 waves, 8 filters, 16 rows per slice
        cases.push_back({0, 237, 20, 32, 100, 5, 1, 1, 1024});   // GFP: 233 positions = 15 M tiles on 16 waves, 59 k-steps (the last one partial) per weight-gradient row
        cases.push_back({0, 26, 20, 32, 16, 7, 4, 2, 512});      // two rows per slice (the window restarts per row), kernel 7 (residue 3 holds one tap + the bias)
    }
    for (const Case& c : cases) {
        const int P = simt_ref_params(c.kind, c.L, c.A, c.F, c.H, c.K);
        const SimtProblem p = simt_problem(c.kind, c.L, c.A, c.F, c.H, c.K, c.rows, c.R, P, 1000u + (unsigned)c.L);
        std::vector<float> part_ref, part;
        const std::vector<float> want = simt_ref_step(p, &part_ref);
        // the device branches as they are (plain rows): the emulator's own check -- fxt_gemm's device side has run on MI355X
        std::vector<float> got;
        const bool only_c32 = quick && c.nthr == 1024 && c.kind == 0 && c.F == 32;     // (sanitizer builds: the sixteen-wave case runs the default form only)
        if (!only_c32) {
            got = emul_step(p, 0, c.nthr, 0, &part);
            bad += same("gradient partials", part_ref, part, p, 0, c.nthr, 0);
            bad += same("weights", want, got, p, 0, c.nthr, 0);
        }
        if (c.kind != 0 || (c.F & (c.F - 1))) continue;
        if (!only_c32) {
            got = emul_step(p, 1, c.nthr, 0, &part);           // rotated rows
            bad += same("gradient partials", part_ref, part, p, 1, c.nthr, 0);
            bad += same("weights", want, got, p, 1, c.nthr, 0);
        }
        const int L1 = c.L - c.K + 1;
        if (c.F == 32 && fxt_conv32_ok(c.R * L1, c.F, c.nthr / 64)) {
            for (int taps : {1, 3, 8}) {                       // the F = 32 form: paired tiles over fragment rows, sliding-window weight gradient
                if (quick && taps != 8) continue;              // (the sanitizer builds: one group size; every group size in the full run)
                if (quick && c.nthr == 1024 && (c.A != 20 || c.H != 100)) continue;
                got = emul_step(p, 3, c.nthr, taps, &part);
                bad += same("gradient partials", part_ref, part, p, 3, c.nthr, taps);
                bad += same("weights", want, got, p, 3, c.nthr, taps);
                if (c.A == 20 && c.H == 100 && c.K == 5 && c.R == 1 && taps != 1 && !(quick && c.nthr == 1024)) {      // k_train_fb_c32p's instantiation: compile-time tap counts, the window renamed in blocks of five k-steps
                    got = emul_step(p, 4, c.nthr, taps, &part);
                    bad += same("gradient partials", part_ref, part, p, 4, c.nthr, taps);
                    bad += same("weights", want, got, p, 4, c.nthr, taps);
                }
            }
        }
        if ((c.F & 31) || !fxt_staged_ok(c.R * L1, c.F, c.F, c.F, c.nthr / 64) || only_c32) continue;
        for (int taps : {1, 2, 6}) {                           // + staged conv kernels, `taps` taps per group
            if (quick && taps != 2) continue;
            got = emul_step(p, 2, c.nthr, taps, &part);
            bad += same("gradient partials", part_ref, part, p, 2, c.nthr, taps);
            bad += same("weights", want, got, p, 2, c.nthr, taps);
        }
    }
    // the canonical instantiations of the BASELINE surrogates (train.hip case numbers), 8 rows per slice, a ragged last slice
    struct Canon { int canon, kind, L, A, F, H, K, rows, nthr; };
    std::vector<Canon> canons = {{4, 0, 8, 4, 32, 100, 5, 11, 256}, {2, 1, 14, 4, 0, 100, 0, 9, 256}};
    if (!quick) {
        canons.push_back({1, 0, 10, 4, 32, 100, 5, 8, 512});
        canons.push_back({3, 2, 25, 20, 0, 100, 0, 13, 256});
        canons.push_back({0, 0, 8, 4, 32, 100, 5, 11, 1024});      // the shape-agnostic code over the same LDS image, 16 waves
        canons.push_back({6, 0, 14, 4, 32, 100, 5, 10, 512});      // RNA length, FOUR rows per slice (round 5: the weights fit beside them)
    }
    for (const Canon& c : canons) {
        const int P = simt_ref_params(c.kind, c.L, c.A, c.F, c.H, c.K);
        const SimtProblem p = simt_problem(c.kind, c.L, c.A, c.F, c.H, c.K, c.rows, c.canon == 6 ? 4 : 8, P, 2000u + (unsigned)c.L);
        std::vector<float> part_ref, part;
        const std::vector<float> want = simt_ref_step(p, &part_ref);
        const std::vector<float> got = emul_step_canon(p, c.canon, c.nthr, &part);
        bad += same("gradient partials (canonical)", part_ref, part, p, 10 + c.canon, c.nthr, 0);
        bad += same("weights (canonical)", want, got, p, 10 + c.canon, c.nthr, 0);
    }
    if (bad) { std::printf("simt_train: %d FAILED\n", bad); return 1; }
    std::printf("simt_train: ok\n");
    return 0;
}
