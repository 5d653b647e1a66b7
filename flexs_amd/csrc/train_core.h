// One mini-batch training step of the reference surrogates (CNN / MLP / GlobalEpistasis) -- forward in training mode,
// MSE loss, reverse-mode gradients, Keras-form Adam -- as two device phases:
//
//   fxt_forward_backward   one workgroup = (member, slice of R mini-batch rows): forward through every layer, backward
//                          through every layer, the slice's gradient of EVERY parameter written to its own row of
//                          `partial` (no atomics: the sum over slices happens in a fixed order in the second phase,
//                          so a fit is deterministic whatever the grid looks like);
//   fxt_adam               one thread per parameter: g = sum over slices (slice order), then
//                          m <- b1 m + (1-b1) g, v <- b2 v + (1-b2) g^2, w <- w - lr_t m / (sqrt(v) + eps).
//
// Replaces `self.model.fit(...)` of flexs/baselines/models/keras_model.py:60-67 for the architectures compiled at
// cnn.py:23-56, mlp.py:21-33, global_epistasis_model.py:26-37 (loss "MSE", optimizer "adam" = tf.keras Adam defaults).
// The arithmetic follows oracle/train_np.py line by line (ties of GlobalMaxPooling1D share the gradient, ReLU has
// gradient 0 at 0, Dropout(0.25) scales the kept units by 1 / 0.75, the loss is the mean over the VALID rows of a
// partial last mini-batch).
//
// Every contraction -- Conv1D as an implicit GEMM over (tap, channel), Dense, and their two transposed forms for the
// weight and the input gradients -- goes through ONE routine, fxt_gemm, whose operands are index functors: on the GPU
// a wave owns 16 x 16 output tiles and feeds v_mfma_f32_16x16x4_f32 (exact f32, the f32 MFMA rate equals the vector
// rate on gfx950 but costs 1 issue slot per 1024 MACs instead of 16), out-of-range elements are zeros, so ANY shape
// the constructors accept trains on the matrix pipe.  The same source compiles for the host (FXT_DEVICE 0: threads
// become loops, the MFMA becomes an fmaf chain): that build is what the CPU test-suite holds to oracle/train_np.py
// (fx_debug_train_step_host), so index arithmetic and gradient algebra are verified without a GPU.
#pragma once
#include <cmath>
#include <cstdint>

// FXT_EMUL (tests/native/simt_train.cpp, host, clang++): the DEVICE branches of this file compiled for the CPU and run by one host
// thread per GPU thread -- the MFMA as a rendezvous of a wave's 64 threads, barriers as pthread barriers, LDS as heap memory, plain
// pointers -- so that what only exists on the device side (tile-to-wave dealing, unrolled k-step groups, accumulators kept across
// staging barriers, the barriers themselves: ThreadSanitizer sees a missing one as a data race) is exercised without a GPU.  The
// emulator supplies the few builtins as macros / functions before including this header.
#if defined(__HIP_DEVICE_COMPILE__) || defined(FXT_EMUL)
#define FXT_DEVICE 1
#else
#define FXT_DEVICE 0
#endif
#if defined(__HIPCC__)
#define FXT_HD __host__ __device__ __forceinline__
#else
#define FXT_HD inline
#endif

#define FXT_MAX_LAYERS 4
#define FXT_DROPOUT 0.25f
#define FXT_LR 1e-3
#define FXT_BETA_1 0.9
#define FXT_BETA_2 0.999
#define FXT_EPSILON 1e-7f

// Static description of one member's network: parameter offsets in Keras get_weights() order.
struct FxtNet {
    int kind, L, A, F, H, K;    // FX_CNN 0 / FX_MLP 1 / FX_GE 2
    int L1, K3;                 // CNN: conv output length L - K + 1, conv3 taps A - 1
    int ldx;                    // CNN: row stride of the F-wide activation arrays of the workspace ("Row strides" below)
    int P;                      // parameter count
    int off_cw[3], off_cb[3];   // CNN: conv kernels (taps, Cin, Cout) and biases
    int nl;                     // dense layers
    int dim[FXT_MAX_LAYERS + 1];// dense stack widths: input, ..., output (= 1)
    int off_w[FXT_MAX_LAYERS], off_b[FXT_MAX_LAYERS];
    int onehot_in;              // 1 = the dense stack reads the flattened one-hot input (MLP / GE), 0 = pooled features (CNN)
    int drop_layer;             // index of the dense layer whose OUTPUT passes through Dropout (-1 = none)
};

// Row strides (round 4).  The MFMA operand fetches are ds_read_b32 -- 32 banks, lanes 0-31 one group (MI355X_MICROARCH.md
// "LDS"): sixteen rows x two k-columns.  With a row stride of 32 floats the sixteen rows of an A operand -- conv activations
// (F = 32 channels) -- or of a transposed-weight B operand sit on ONE bank: 16-way conflicts, 32 LDS cycles per fetch instead
// of 2, times the waves sharing the CU's LDS; that, not the MFMAs, set the time of the conv phases.  Activation rows are
// therefore F + 2 floats apart (stride / 2 odd: sixteen rows on sixteen distinct even banks, the second k-column on the odd
// ones), and the conv kernels' rows in the LDS image of the weights F + 4 (16-byte rows for the staging copy; 2-way at most).
// The pad columns are never read.  Global weights, gradients and Adam moments keep the Keras layout.  (A host that finds the
// padded workspace just too large for LDS may set FxtNet::ldx back to F: the unpadded workspace in LDS beats the padded one
// in global memory.)
FXT_HD int fxt_ld_x(int F) { return (F & 3) == 0 ? F + 2 : F; }
FXT_HD int fxt_ld_w(int F) { return (F & 7) == 0 ? F + 4 : F; }
// Rotated rows (SWZ; prepared at the end of round 4 for the long protein CNNs, engine option `train_swizzle`, not yet measured).  A
// sequence of 237 residues leaves no room for padded rows: five position-major arrays of 233 x 32 floats are 146 of the 150 KiB, so
// those fits run with unpadded rows (ldx = F) -- every conv operand fetch 16-way conflicted again.  Instead of widening the rows,
// channel c of position-row `row` is stored at column (c + 2 row) mod F (F a power of two >= 32, row stride = F): the sixteen rows
// of an A-operand fetch sit on sixteen distinct even banks and the second k-column on the odd ones, exactly as with F + 2 padding,
// in the same 32 floats.  Only the five position-major arrays (conv outputs and their gradients) are rotated; they are reached
// through fxt_xi and the ...Swz operand functors below, and the unrotated code keeps its own functors (the kernels every other
// fit runs are unchanged instruction for instruction).  Where a value is stored does not change the value: the same bits as the
// unrotated layout (sanitizer driver, CPU).
// LAY 0: rows as they are; 1 (true): rotated rows; 2: FRAGMENT rows (MODE 3, 32 channels, round 5) -- channel c = 16 h + 4 u + kq of a
// row is stored at column 4 ((4 h + kq + (row & 6)) mod 8) + u: the four channels a lane feeds to the four k-steps of a half-tap (same
// kq, u = 0 .. 3) are 16 contiguous bytes, so an A operand is ONE ds_read_b128 per half-tap instead of four ds_read_b32 -- a lone wave
// gets a fifth of the LDS rate on 4-byte reads (MI355X_MICROARCH.md, LDS), and with the matrix pipe handed to the oldest wave the
// conv products ran one wave per SIMD at a time, each waiting on its own reads (round 5, per-wave stamps) -- and the chunk rotation
// by (row & 6) keeps the sixteen lanes of each ds_read_b128 lane group on sixteen distinct 16-byte bank groups for ANY starting row.
template <int LAY>
FXT_HD int fxt_xi(int row, int c, int ld) {
    if (LAY == 2) return row * 32 + 4 * ((((c >> 4) << 2) + (c & 3) + (row & 6)) & 7) + ((c >> 2) & 3);
    return row * ld + (LAY ? ((c + 2 * row) & (ld - 1)) : c);
}
template <bool SWZ, class A, class B> struct FxtPick { typedef A T; };
template <class A, class B> struct FxtPick<true, A, B> { typedef B T; };
// MODE of fxt_forward_backward: 3 = MODE 2 with the F = 32 conv products of "MODE 3" below (paired tiles over conflict-free rotated
// kernel rows, register-prefetched staging, sliding-window weight gradient); 0 = rows as they are (padded or not), 1 = rotated rows, 2 = rotated rows + the gradient array dzA
// over a[2] (fxt_ws alias_dz) + the conv kernels of conv2 / conv3 STAGED through the LDS that frees, a group of taps at a time
// (fxt_gemm_staged): with 165 KiB of weights in global memory every B operand of the protein CNNs' conv products is an L2 round
// trip, eight in flight per wave; staged, a tap's 32 x 32 block is fetched once per workgroup and product instead of once per
// output tile.  The taps are walked in the same order with the same accumulators: the same bits.  (Also prepared at the end of
// round 4 and not yet measured: `train_swizzle` = 2.)

FXT_HD FxtNet fxt_net(int kind, int L, int A, int F, int H, int K) {
    FxtNet n{};
    n.kind = kind; n.L = L; n.A = A; n.F = F; n.H = H; n.K = K;
    int off = 0;
    if (kind == 0) {
        n.L1 = L - K + 1; n.K3 = A - 1; n.ldx = fxt_ld_x(F);
        const int taps[3] = {K, K, A - 1}, cin[3] = {A, F, F};
        for (int i = 0; i < 3; ++i) {
            n.off_cw[i] = off; off += taps[i] * cin[i] * F;
            n.off_cb[i] = off; off += F;
        }
        n.nl = 3; n.dim[0] = F; n.dim[1] = H; n.dim[2] = H; n.dim[3] = 1;
        n.onehot_in = 0; n.drop_layer = 1;
    } else if (kind == 1) {
        n.nl = 4; n.dim[0] = L * A; n.dim[1] = H; n.dim[2] = H; n.dim[3] = H; n.dim[4] = 1;
        n.onehot_in = 1; n.drop_layer = -1;
    } else {
        n.nl = 4; n.dim[0] = L * A; n.dim[1] = 1; n.dim[2] = H; n.dim[3] = H; n.dim[4] = 1;
        n.onehot_in = 1; n.drop_layer = -1;
    }
    for (int i = 0; i < n.nl; ++i) {
        n.off_w[i] = off; off += n.dim[i] * n.dim[i + 1];
        n.off_b[i] = off; off += n.dim[i + 1];
    }
    n.P = off;
    return n;
}

// Where a member's parameters sit in the image the step reads them from: the Keras get_weights() order of FxtNet (global
// memory: `padded` = false) or the LDS image with padded conv-kernel rows.
struct FxtLay {
    int cw[3], cb[3];
    int w[FXT_MAX_LAYERS], b[FXT_MAX_LAYERS];
    int ldw;                    // row stride of the conv kernels
    int fsh;                    // log2(F) when F is a power of two (the staging copy's row index is a shift then), else -1
    int total;
};
FXT_HD FxtLay fxt_lay(const FxtNet& n, bool padded) {
    FxtLay y{};
    y.ldw = (padded && n.kind == 0) ? fxt_ld_w(n.F) : n.F;
    y.fsh = -1;
    for (int b = 0; b < 16; ++b) if (n.F == (1 << b)) y.fsh = b;
    int off = 0;
    if (n.kind == 0) {
        const int rows[3] = {n.K * n.A, n.K * n.F, n.K3 * n.F};
        for (int i = 0; i < 3; ++i) {
            y.cw[i] = off; off += rows[i] * y.ldw;
            y.cb[i] = off; off += n.F;
        }
    }
#if FXT_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < FXT_MAX_LAYERS; ++i) {
        y.w[i] = off; if (i < n.nl) off += n.dim[i] * n.dim[i + 1];
        y.b[i] = off; if (i < n.nl) off += n.dim[i + 1];
    }
    y.total = off;
    return y;
}
// image offset of parameter `g` (Keras order).  Everything behind a conv kernel keeps its distance to that kernel's end.
FXT_HD int fxt_image_off(const FxtNet& n, const FxtLay& y, int g) {
    if (y.ldw == n.F || n.kind != 0) return g;
#if FXT_DEVICE
#pragma unroll
#endif
    for (int c = 2; c >= 0; --c) {
        if (g >= n.off_cw[c]) {
            const int d = g - n.off_cw[c], size = n.off_cb[c] - n.off_cw[c];
            // (a hardware integer division is ~40 instructions, and this runs per 16-byte piece of the 90 KiB image, every step)
            if (d < size) return y.cw[c] + d + (y.fsh >= 0 ? d >> y.fsh : d / n.F) * (y.ldw - n.F);
            return y.cb[c] + (d - size);
        }
    }
    return g;
}

// Workspace of one slice (floats): codes, activations, gradients.  Offsets relative to the slice's base.
struct FxtWs {
    int codes;                  // R x L      alphabet indices (stored as int32 in the float buffer)
    int ylab, yvalid;           // R          the rows' labels and validity (1.0 / 0.0), fetched together with the codes
    int ldF;                    // row stride of the F-wide arrays (fxt_ld_x)
    int a[3];                   // R x L1 x ldF post-ReLU conv outputs
    int dzA, dzB;               // R x L1 x ldF gradient ping-pong
    int g, cnt, dg;             // R x ldF     pooled maxima, tie counts, gradient
    int zero;                   // (alias_dz layouts) ldF zeros: where MODE 3's operand fetches of positions outside a row are sent (no select on the value)
    int act[FXT_MAX_LAYERS];    // R x dim[i+1] post-activation (post-dropout) outputs
    int du[FXT_MAX_LAYERS];     // R x dim[i+1] gradient w.r.t. the pre-activation
    int total;
};

// alias_dz: the gradient array dzA lies OVER the last conv output a[2] (the max-pool backward turns one into the other element by
// element, and nothing reads a[2] afterwards): four position-major arrays instead of five (MODE 2 below).
FXT_HD FxtWs fxt_ws(const FxtNet& n, int R, bool alias_dz = false) {
    FxtWs w{};
    int off = 0;
    w.codes = off; off += R * n.L;
    w.ylab = off; off += R;
    w.yvalid = off; off += R;
    if (n.kind == 0) {
        w.ldF = n.ldx;
        if (alias_dz) off = (off + 3) & ~3;     // (16-byte aligned arrays: MODE 3 fetches operands with ds_read_b128)
        const int s = R * n.L1 * w.ldF;
        for (int i = 0; i < 3; ++i) { w.a[i] = off; off += s; }
        if (alias_dz) w.dzA = w.a[2];
        else { w.dzA = off; off += s; }
        w.dzB = off; off += s;
        w.g = off; off += R * w.ldF;
        w.cnt = off; off += R * w.ldF;
        w.dg = off; off += R * w.ldF;
        w.zero = off; if (alias_dz) off += w.ldF;
    }
    // (fixed trip counts: with run-time bounds the offset table is indexed dynamically and lives in scratch memory)
#if FXT_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < FXT_MAX_LAYERS; ++i) { w.act[i] = off; if (i < n.nl) off += R * n.dim[i + 1]; }
#if FXT_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < FXT_MAX_LAYERS; ++i) { w.du[i] = off; if (i < n.nl) off += R * n.dim[i + 1]; }
    w.total = (off + 3) & ~3;
    return w;
}

// One member's training job as the kernels see it.
struct FxtJob {
    FxtNet net;
    int batch;                  // mini-batch slots per step
    int steps_per_epoch, total_steps;
    int n;                      // data-set rows
    int R, S;                   // rows per slice, slices per step = ceil(batch / R)
    float* w;                   // [P]
    float* adam_m; float* adam_v;
    float* partial;             // [S][pstride]: gradient partials, then the slice's sum of squared errors at [P]
    int pstride;                // floats between two slices' rows (0 = P + 1).  fx_train_fit rounds it up to 32 floats: rows of P + 1 = 41 374 floats start 120 bytes into a cache line, so every 64-byte run of gradient stores straddled two sectors and every 256-byte run the Adam kernel reads three lines
    const int32_t* order;       // [total_steps][batch] data-set row per slot, -1 = padding slot
    const uint8_t* keep;        // optional explicit dropout keep mask [total_steps][batch][H]
    unsigned long long seed;    // in-kernel dropout stream when keep == nullptr
    const float* lr_t;          // [total_steps] lr sqrt(1 - b2^t) / (1 - b1^t), t = t0 + step + 1 (host, double precision)
    float* ws; long long ws_slice;   // workspace, floats per slice
    int ws_in_lds;              // 1 = the slice's workspace fits the workgroup's LDS and lives there
    int w_in_lds;               // 1 = ... and the member's weights fit next to it (staged at kernel start)
    int agent_io;               // 1 = gradient partials and updated weights are written through / read past the non-coherent cache levels (the one-launch fit: workgroups exchange them inside a launch)
    int split_off;              // > 0: offset (floats) of the split-K scratch in the workgroup's LDS, 0 = products are not cut along the contraction
    int canon;                  // > 0: a canonical shape with its own compile-time instantiation (train.hip), 0 = the shape-agnostic code, -1 = the shape-agnostic code over rotated rows (SWZ)
    float* step_loss;           // [total_steps] mean squared error of the step's valid rows (before the update)
    unsigned long long* dbg;    // profiling aid (engine option "train_trace"): phase timestamps of workgroup (0, 0), else nullptr
};

// Stores / loads that are coherent across the XCDs without fences (agent scope, `sc1`): what the one-launch fit hands from
// workgroup to workgroup -- gradient partials, updated weights -- inside a launch (see mfma_common.h fx_store16_agent for the
// why: a release / acquire pair is a write-back + invalidate of the XCD's whole L2, ~0.5 us, serialised per XCD).
#if defined(FXT_EMUL)
inline void fxt_store_agent(float* p, float v) { *p = v; }
inline void fxt_load8_agent(const float* p, long long stride, float (&v)[8]) { for (int k = 0; k < 8; ++k) v[k] = p[k * stride]; }
#elif FXT_DEVICE
__device__ __forceinline__ void fxt_store_agent(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void fxt_load8_agent(const float* p, long long stride, float (&v)[8]) {   // v[k] = p[k * stride], eight in flight
    asm volatile("global_load_dword %0, %8, off sc1\n\tglobal_load_dword %1, %9, off sc1\n\tglobal_load_dword %2, %10, off sc1\n\t"
                 "global_load_dword %3, %11, off sc1\n\tglobal_load_dword %4, %12, off sc1\n\tglobal_load_dword %5, %13, off sc1\n\t"
                 "global_load_dword %6, %14, off sc1\n\tglobal_load_dword %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(p + stride), "v"(p + 2 * stride), "v"(p + 3 * stride), "v"(p + 4 * stride), "v"(p + 5 * stride),
                   "v"(p + 6 * stride), "v"(p + 7 * stride)
                 : "memory");
}
#endif

// nothing is scheduled across this point (device); the emulator and the host have no scheduler to restrain
#if FXT_DEVICE && !defined(FXT_EMUL)
#define FXT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FXT_SCHED_FENCE() ((void)0)
#endif
// q = x / d for 0 <= x < 2^16, 1 <= d < 2^16 without a hardware divide (an integer division is ~40 instructions on
// the GPU and the GEMM tiles decompose their row index once each): q = (x * ceil(2^32 / d)) >> 32, exact in that range.
struct FxtDiv { unsigned d, magic; };
FXT_HD FxtDiv fxt_div(int d) { return FxtDiv{(unsigned)d, d > 1 ? (unsigned)(0xFFFFFFFFull / (unsigned)d + 1ull) : 0u}; }
FXT_HD int fxt_quot(int x, FxtDiv dv) {
    return dv.d > 1 ? (int)(((unsigned long long)(unsigned)x * dv.magic) >> 32) : x;
}

// Execution context of a workgroup: on the device one instance per thread, on the host ONE instance that plays
// every thread in turn (a phase is data-parallel; phases are separated by fxt_sync).
struct FxtWg { int tid, nthr; };
// The wave's index as a SCALAR: tid >> 6 is the same in all 64 lanes, but the compiler cannot know it -- a tile loop dealt by
// `tid >> 6` runs under exec masking with vector address arithmetic, an `if (wave-owned job)` becomes a divergent branch around every
// MFMA (round 5: the weight-gradient loop's MFMAs each sat behind a saveexec / branch pair).  readfirstlane hands it over in an SGPR.
FXT_HD int fxt_wave(const FxtWg& wg) {
#if FXT_DEVICE && !defined(FXT_EMUL)
    return __builtin_amdgcn_readfirstlane(wg.tid >> 6);
#else
    return wg.tid >> 6;
#endif
}
// first dense layer on a one-hot input, weight gradient: from this many input rows (L x A + 1) on, one thread per element instead of MFMA tiles of two k-steps (train_step.h)
#define FXT_ONEHOT_WGRAD_DIRECT 256
#define FXT_FOR(i, count, wg) for (int i = (wg).tid; i < (count); i += (wg).nthr)
FXT_HD void fxt_sync() {
#if FXT_DEVICE
    __syncthreads();
#endif
}
// Barrier between two phases of fxt_forward_backward.  When the slice's workspace lives in LDS (WSAS = 3) everything one phase
// hands to the next is in LDS, and what a phase writes to GLOBAL memory -- its gradient partials -- is read by the next LAUNCH
// only: the barrier then orders the LDS traffic alone (release / acquire fences restricted to the local address space), and the
// partial stores of six backward phases drain behind the following phases instead of being waited for (~1 us of L2 round trip)
// at each of them, which is what a full __syncthreads() -- a workgroup fence over every address space -- costs.
template <int WSAS>
FXT_HD void fxt_sync_ws() {
#if FXT_DEVICE
    if constexpr (WSAS == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
        __syncthreads();
    }
#endif
}
// phase stamp k of workgroup (0, 0): the 100 MHz wall clock after the phase's barrier
#if FXT_DEVICE
#define FXT_STAMP(k) do { if (j.dbg && wg.tid == 0 && slice == 0) j.dbg[(k)] = wall_clock64(); } while (0)
#else
#define FXT_STAMP(k) do { } while (0)
#endif

FXT_HD long long fxt_pstride(const FxtJob& j) { return j.pstride ? j.pstride : j.net.P + 1; }
FXT_HD bool fxt_keep(const FxtJob& j, int step, int slot, int h) {
    if (slot >= j.batch) return true;                    // a slot past the mini-batch (the last slice's overhang): no mask entry, no gradient
    if (j.keep) return j.keep[((long long)step * j.batch + slot) * j.net.H + h] != 0;
    unsigned long long z = j.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(((long long)step * j.batch + slot) * j.net.H + h + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) < 3221225472u;            // P(keep) = 0.75 = 1 - FXT_DROPOUT
}

// The contraction routine, its operand functors and the slice step live in their own files (round 6: no source file over ~900 lines);
// same translation unit, same order as before.
#include "train_gemm.h"
#include "train_operands.h"
#include "train_step.h"

// Sum of the slices' partial gradients (slice order) + one Keras-Adam update of parameter i.
FXT_HD void fxt_adam(const FxtJob& j, int step, int i) {
    const int S = j.S;
    const long long ps = fxt_pstride(j);
    const float* part = j.partial + i;
    float gsum = 0.f;
#if FXT_DEVICE
    if (j.agent_io) {
        // the one-launch fit: the partials were written through by the other workgroups of the member (fxt_store_agent) and are
        // read past the non-coherent cache levels, eight at a time; the new weight is written through for the same reason.
        // Same sum (slice order), same update: same bits.
        for (int s0 = 0; s0 < S; s0 += 8) {
            float v[8];
            fxt_load8_agent(part + (long long)s0 * ps, (s0 + 8 <= S) ? ps : 0ll, v);   // (a ragged tail re-reads its first slice: handled below)
            if (s0 + 8 <= S) {
#pragma unroll
                for (int u = 0; u < 8; ++u) gsum += v[u];
            } else {
                for (int u = 0; s0 + u < S; ++u) gsum += __hip_atomic_load(part + (long long)(s0 + u) * ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const float b1 = (float)FXT_BETA_1, b2 = (float)FXT_BETA_2;
        const float m = b1 * j.adam_m[i] + (1.f - b1) * gsum;
        const float v = b2 * j.adam_v[i] + (1.f - b2) * gsum * gsum;
        j.adam_m[i] = m;
        j.adam_v[i] = v;
        const float w_old = __hip_atomic_load(&j.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fxt_store_agent(&j.w[i], w_old - j.lr_t[step] * m / (sqrtf(v) + FXT_EPSILON));
        return;
    }
#endif
    for (int s0 = 0; s0 < S; s0 += 32) {     // slice order; 32 loads in flight -- one L2 round trip for the 32 slices of a 256-row batch -- (x + 0.f leaves x as it is)
        float v[32];
#if FXT_DEVICE
#pragma unroll
#endif
        for (int u = 0; u < 32; ++u) v[u] = (s0 + u < S) ? part[(long long)(s0 + u) * ps] : 0.f;
#if FXT_DEVICE
#pragma unroll
#endif
        for (int u = 0; u < 32; ++u) gsum += v[u];
    }
    const float b1 = (float)FXT_BETA_1, b2 = (float)FXT_BETA_2;
    const float m = b1 * j.adam_m[i] + (1.f - b1) * gsum;
    const float v = b2 * j.adam_v[i] + (1.f - b2) * gsum * gsum;
    j.adam_m[i] = m;
    j.adam_v[i] = v;
    j.w[i] = j.w[i] - j.lr_t[step] * m / (sqrtf(v) + FXT_EPSILON);
}

// The same update for parameters i4 ... i4 + 3 at once (device; rows of the partial array 16-byte aligned: FxtJob::pstride a multiple
// of four, i4 a multiple of four, i4 + 3 < P).  A dword load per lane is 256 bytes per wave instruction and the kernel was bound by
// the rate of those instructions, not by memory (round 5: 127 MB of partials of a GFP-length step in 77 us = 1.7 TB/s whatever the
// number of loads in flight); 16 bytes per lane quarters them.  Every parameter's additions in the same (slice) order: the same bits.
#if defined(__HIPCC__)
__device__ __forceinline__ void fxt_adam4(const FxtJob& j, int step, int i4) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int S = j.S;
    const long long ps = fxt_pstride(j);
    const float* part = j.partial + i4;
    f4_t g = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 24) {
        f4_t v[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) v[u] = (s0 + u < S) ? *reinterpret_cast<const f4_t*>(part + (long long)(s0 + u) * ps) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 24; ++u) { g[0] += v[u][0]; g[1] += v[u][1]; g[2] += v[u][2]; g[3] += v[u][3]; }
    }
    const float b1 = (float)FXT_BETA_1, b2 = (float)FXT_BETA_2, lr = j.lr_t[step];
    f4_t m = *reinterpret_cast<const f4_t*>(j.adam_m + i4), v = *reinterpret_cast<const f4_t*>(j.adam_v + i4), w = *reinterpret_cast<const f4_t*>(j.w + i4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m[c] = b1 * m[c] + (1.f - b1) * g[c];
        v[c] = b2 * v[c] + (1.f - b2) * g[c] * g[c];
        w[c] = w[c] - lr * m[c] / (sqrtf(v[c]) + FXT_EPSILON);
    }
    *reinterpret_cast<f4_t*>(j.adam_m + i4) = m;
    *reinterpret_cast<f4_t*>(j.adam_v + i4) = v;
    *reinterpret_cast<f4_t*>(j.w + i4) = w;
}
#endif

FXT_HD void fxt_step_loss(const FxtJob& j, int step) {
    const int sidx = step % j.steps_per_epoch;
    const int nvalid = (j.n - sidx * j.batch) < j.batch ? (j.n - sidx * j.batch) : j.batch;
    float sse = 0.f;
#if FXT_DEVICE
    if (j.agent_io) {
        for (int s = 0; s < j.S; ++s) sse += __hip_atomic_load(&j.partial[(long long)s * fxt_pstride(j) + j.net.P], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
#endif
    for (int s = 0; s < j.S; ++s) sse += j.partial[(long long)s * fxt_pstride(j) + j.net.P];
    j.step_loss[step] = sse / (float)nvalid;
}
