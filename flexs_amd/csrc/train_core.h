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

// ---------------------------------------------------------------------------------------------------------------
// C[m][n] = sum over (ko, ki) of A(m, ko, ki) * B(ko, ki, n),  m < Md, n < Nd, ko < Ko, ki < Ki.
// The contraction index is kept as a PAIR so that conv taps / batch rows never need a division in the inner loop;
// each ko runs ceil(Ki / 4) k-steps (the overhang multiplies zeros).  FA: prep(m, kq) -> per-lane state (once per
// tile), at(state, ko, k0) = A(m, ko, k0 + kq); FB: prep(n, kq), at(state, ko, k0) = B(ko, k0 + kq, n); FC: put(m, n, value).
// `first_wave`: the wave that takes tile 0 (tiles go round-robin from there).  Two products of one phase -- the few long tiles of
// an input gradient and the many short ones of a weight gradient -- are dealt as ONE sequence: the second call passes
// fxt_tiles(first product) as its first_wave, so no wave gets a long tile on top of a full share of short ones.
FXT_HD int fxt_tiles(int Md, int Nd) { return ((Md + 15) >> 4) * ((Nd + 15) >> 4); }
// `split` (device, LDS scratch of FXT_SPLIT_FLOATS floats, or null): a product with FEW tiles and a LONG contraction -- conv2
// forward: 4 tiles of 40 k-steps for 16 waves -- is cut along the contraction as well: the groups of eight k-steps of a tile are
// dealt to up to four waves, the partial tiles meet in the scratch (one LDS-only barrier), and the first wave of a tile adds them
// in split order and runs the epilogue.  How a product is cut depends on its shape and the workgroup size only
// (fxt_split_ways), so every instantiation of this source -- shape-agnostic or canonical -- sums in the same order.
// ALL waves of the workgroup must call fxt_gemm together when `split` is given (the barrier).
#define FXT_SPLIT_FLOATS 4096
FXT_HD int fxt_split_ways(int tiles, int Ko, int Ki, int nw) {
    if (Ki < 32 || tiles * 2 > nw) return 1;
    const int groups = Ko * ((Ki + 31) >> 5);
    int ways = nw / tiles;
    if (ways > 4) ways = 4;
    if (ways > groups) ways = groups;
    return ways < 1 ? 1 : ways;
}
// workgroup jobs of a product (tiles x the ways it is cut): what the NEXT product of the phase passes as its first_wave
FXT_HD int fxt_jobs(int Md, int Nd, int Ko, int Ki, int nw, bool can_split) {
    const int t = fxt_tiles(Md, Nd);
#if defined(FX_AB)
    return t * (can_split ? fxt_split_ways(t, Ko, Ki, nw) : 1);
#else
    (void)Ko; (void)Ki; (void)nw; (void)can_split;
    return t;
#endif
}
template <class FA, class FB, class FC, class SP = float*>
FXT_HD void fxt_gemm(const FxtWg& wg, int Md, int Nd, int Ko, int Ki, const FA& fa, const FB& fb, const FC& fc, int first_wave = 0,
                     SP split = nullptr) {
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8;                     // k-steps whose operand loads are in flight together
    const int lane = wg.tid & 63, nw = wg.nthr >> 6;
    const int wave = ((wg.tid >> 6) + nw - first_wave % nw) % nw;
    const int i = lane & 15, kq = lane >> 4;
    const int tn = (Nd + 15) >> 4, tiles = ((Md + 15) >> 4) * tn;
    const int T = Ko * ((Ki + 3) >> 2);      // k-steps of the whole contraction, (ko, k0) in row-major order
    // (measured: the cut LOSES -- 24.6 -> 33.1 us per forward+backward launch of the 3 x CNN step, every phase it touches
    //  slower: the extra barrier and the partial tiles through LDS cost more than the idle waves were worth,
    //  profiles/r4_train_split_ab.log -- so it is compiled into the A/B build only; elsewhere ways == 1 folds it all away)
#if defined(FX_AB)
    const int ways = split ? fxt_split_ways(tiles, Ko, Ki, nw) : 1;
#else
    constexpr int ways = 1;
    (void)split;
#endif
    const int gpk = (Ki + 31) >> 5, groups = Ko * gpk;
    auto epilogue = [&](int t, f4_t acc) {
        const int m0 = (t / tn) << 4, n = ((t % tn) << 4) + i;
        if (n < Nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = m0 + 4 * kq + r;
                if (mr < Md) fc.put(mr, n, acc[r]);
            }
        }
    };
    for (int job = wave; job < tiles * ways; job += nw) {
        const int t = ways > 1 ? job / ways : job, sp = ways > 1 ? job - t * ways : 0;
        const int g_lo = groups * sp / ways, g_hi = groups * (sp + 1) / ways;      // this job's groups of eight k-steps (ways > 1)
        const int m0 = (t / tn) << 4, n0 = (t % tn) << 4;
        const int m = m0 + i, n = n0 + i;
        const bool mok = m < Md, nok = n < Nd;
        const auto sa = fa.prep(mok ? m : 0, kq);
        const auto sb = fb.prep(nok ? n : 0, kq);
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        // The operands come from L2 / LDS through index functors: issued one k-step at a time every MFMA would wait a
        // full memory round trip (the first build ran at ~1 us per k-step).  U k-steps are loaded first, then multiplied.
        // Rows past Md / columns past Nd need no masking: row i of A only reaches row i of the product, column j of B only
        // column j, and those are never stored (their lanes read row / column 0).  Only the contraction index must be
        // exact: a k-step past Ki has to contribute zero.
        if (Ki >= 4 * U) {
            // long inner index (conv taps x channels, dense layers): whole groups of U k-steps inside one `ko` need no
            // range logic at all -- (ko, k0) are wave-uniform, the U fetches differ by constant offsets; the functors'
            // own checks (conv positions) depend on ko only
            for (int ko = 0; ko < Ko; ++ko) {
                int k0 = 0;
                for (; k0 + 4 * U <= Ki; k0 += 4 * U) {
                    if (ways > 1) { const int gi = ko * gpk + (k0 >> 5); if (gi < g_lo || gi >= g_hi) continue; }
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { a[u] = fa.at(sa, ko, k0 + 4 * u); b[u] = fb.at(sb, ko, k0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
                }
                if (k0 < Ki && (ways == 1 || (ko * gpk + (k0 >> 5) >= g_lo && ko * gpk + (k0 >> 5) < g_hi))) {   // the row's last, partial group
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int kk = k0 + 4 * u;
                        const bool live = kk < Ki;           // wave-uniform
                        const bool kok = kk + kq < Ki;
#if defined(FXT_EMUL)     // (the emulator does not perform a load whose value is masked away: on the device it reads -- and drops -- a
                          //  neighbouring array's element, which a race detector reports and an exact-size buffer cannot hold)
                        a[u] = kok ? fa.at(sa, ko, kk) : 0.f;
                        b[u] = kok ? fb.at(sb, ko, kk) : 0.f;
#else
                        const float av = fa.at(sa, ko, live ? kk : 0), bv = fb.at(sb, ko, live ? kk : 0);
                        a[u] = kok ? av : 0.f;
                        b[u] = kok ? bv : 0.f;
#endif
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (k0 + 4 * u < Ki) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
                }
            }
        } else if (Ki <= 4) {
            // one k-step per `ko` (conv weight gradients of short sequences: the four positions of a row): k-step = ko
            for (int s = 0; s < Ko; s += U) {
                float a[U], b[U];
                const bool kin = kq < Ki;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = s + u < Ko;            // wave-uniform
#if defined(FXT_EMUL)
                    a[u] = (live && kin) ? fa.at(sa, s + u, 0) : 0.f;
                    b[u] = (live && kin) ? fb.at(sb, s + u, 0) : 0.f;
#else
                    const float av = fa.at(sa, live ? s + u : 0, 0), bv = fb.at(sb, live ? s + u : 0, 0);
                    a[u] = (live && kin) ? av : 0.f;
                    b[u] = (live && kin) ? bv : 0.f;
#endif
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s + u < Ko) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
        } else {
            // short inner index (weight gradients: positions of a row, rows of a slice): the (ko, k0) pairs are walked as
            // one flat sequence of k-steps, wave-uniform counters instead of a division
            int ko = 0, k0 = 0;
            for (int s = 0; s < T; s += U) {
                float a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = s + u < T;
                    const bool kok = live && k0 + kq < Ki;
#if defined(FXT_EMUL)
                    a[u] = kok ? fa.at(sa, ko, k0) : 0.f;
                    b[u] = kok ? fb.at(sb, ko, k0) : 0.f;
#else
                    const float av = fa.at(sa, live ? ko : 0, live ? k0 : 0);
                    const float bv = fb.at(sb, live ? ko : 0, live ? k0 : 0);
                    a[u] = kok ? av : 0.f;
                    b[u] = kok ? bv : 0.f;
#endif
                    k0 += 4;
                    if (k0 >= Ki) { k0 = 0; ++ko; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s + u < T) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
        }
        if (ways > 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) split[(job * 64 + lane) * 4 + r] = acc[r];
        } else {
            epilogue(t, acc);
        }
    }
    if (ways > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        for (int t = wave; t < tiles; t += nw) {
            f4_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = split[((t * ways) * 64 + lane) * 4 + r];
            for (int sp = 1; sp < ways; ++sp)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += split[((t * ways + sp) * 64 + lane) * 4 + r];
            epilogue(t, acc);
        }
    }
#else
    (void)wg; (void)first_wave; (void)split;
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < Nd; ++n) {
            float acc = 0.f;
            for (int ko = 0; ko < Ko; ++ko)
                for (int ki = 0; ki < Ki; ++ki)          // element ki belongs to lane group kq = ki % 4 of k-step k0 = ki - kq
                    acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko, ki & ~3), acc);
            fc.put(m, n, acc);
        }
#endif
}

// ---- address spaces ---------------------------------------------------------------------------------------------
// A pointer whose address space the compiler does not know is read with flat_load, and a FLAT access that resolves to
// LDS is several times slower than ds_read (phase timeline, profiles/r3_train_trace.log: ~1.1 us per group of eight
// k-steps with every operand in LDS).  The step is therefore compiled per placement -- workspace in LDS (3) or global
// memory (1), weights in LDS or global memory -- with address-space-qualified pointer types; the host build has one.
#if defined(FXT_EMUL)
typedef float fxt_f4 __attribute__((ext_vector_type(4), aligned(4)));   // (host memory of the emulator: no 16-byte promise)
template <int AS> struct FxtMem { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const fxt_f4* CF4; typedef fxt_f4* F4; };
#elif FXT_DEVICE
typedef float fxt_f4 __attribute__((ext_vector_type(4)));
template <int AS> struct FxtMem {
    typedef __attribute__((address_space(AS))) float* F;
    typedef const __attribute__((address_space(AS))) float* CF;
    typedef __attribute__((address_space(AS))) int* I;
    typedef const __attribute__((address_space(AS))) int* CI;
    typedef const __attribute__((address_space(AS))) fxt_f4* CF4;      // sixteen bytes at a time (fxt_gemm_staged)
    typedef __attribute__((address_space(AS))) fxt_f4* F4;
};
template <> struct FxtMem<0> { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const fxt_f4* CF4; typedef fxt_f4* F4; };   // (flat / host)
#else
template <int AS> struct FxtMem { typedef float* F; typedef const float* CF; typedef int* I; typedef const int* CI; typedef const float* CF4; };   // (CF4: never dereferenced by the host build)
#endif


// fxt_gemm for a conv product whose B operand -- taps [0, Ko) of a conv kernel, `rows_per_tap` rows of F floats each in global
// memory at `wsrc` -- is staged through `wbuf` (LDS on the device; rows `ldw` floats apart) in groups of G taps.  `fb` is the B functor
// built OVER wbuf: it is handed the tap index relative to its group.  Ki is a multiple of 32 (whole groups of eight k-steps), a wave
// owns at most FXT_STAGED_TPW tiles whose accumulators stay in registers across the groups (the caller checks both: fxt_staged_ok).
// ALL threads of the workgroup call it together (two LDS barriers per group).
#define FXT_STAGED_TPW 2
FXT_HD bool fxt_staged_ok(int Md, int Nd, int Ki, int F, int nw) { return (Ki & 31) == 0 && (F & 3) == 0 && fxt_tiles(Md, Nd) <= nw * FXT_STAGED_TPW; }
template <int WSAS, int WAS, class FA, class FB, class FC>
FXT_HD void fxt_gemm_staged(const FxtWg& wg, int Md, int Nd, int Ko, int Ki, const FA& fa, const FB& fb, const FC& fc,
                            typename FxtMem<WAS>::CF wsrc, typename FxtMem<WSAS>::F wbuf, int G, int rows_per_tap, int F, int ldw) {
    if (G < 1) G = Ko;                                     // (never from the host's sizing; a group must advance)
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8, TPW = FXT_STAGED_TPW;
    const int lane = wg.tid & 63, nw = wg.nthr >> 6, wave = wg.tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int tn = (Nd + 15) >> 4, tiles = ((Md + 15) >> 4) * tn;
    f4_t acc[TPW];
    decltype(fa.prep(0, 0)) sa[TPW];
    decltype(fb.prep(0, 0)) sb[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * nw;
        const int m = ((t / tn) << 4) + i, n = ((t % tn) << 4) + i;
        acc[q] = f4_t{0.f, 0.f, 0.f, 0.f};
        sa[q] = fa.prep((t < tiles && m < Md) ? m : 0, kq);
        sb[q] = fb.prep((t < tiles && n < Nd) ? n : 0, kq);
    }
    const int f4_per_row = F >> 2;
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        fxt_sync_ws<WSAS>();                               // everybody is through with the previous group's taps (or the previous phase)
        const int pieces = (g1 - g0) * rows_per_tap * f4_per_row;
        for (int p = wg.tid; p < pieces; p += wg.nthr) {
            const int row = p / f4_per_row, c4 = p - row * f4_per_row;
            const f4_t v = *(typename FxtMem<WAS>::CF4)(wsrc + ((g0 * rows_per_tap + row) * F + 4 * c4));
            const int o = row * ldw + 4 * c4;               // (scalar stores: they keep wbuf's address space -- ds_write, not flat)
            wbuf[o] = v[0]; wbuf[o + 1] = v[1]; wbuf[o + 2] = v[2]; wbuf[o + 3] = v[3];
        }
        fxt_sync_ws<WSAS>();
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (wave + q * nw >= tiles) continue;          // (wave-uniform)
            for (int ko = g0; ko < g1; ++ko) {
                for (int k0 = 0; k0 < Ki; k0 += 4 * U) {
                    float a[U], b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { a[u] = fa.at(sa[q], ko, k0 + 4 * u); b[u] = fb.at(sb[q], ko - g0, k0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[q], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * nw;
        if (t >= tiles) continue;
        const int m0 = (t / tn) << 4, n = ((t % tn) << 4) + i;
        if (n < Nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = m0 + 4 * kq + r;
                if (mr < Md) fc.put(mr, n, acc[q][r]);
            }
        }
    }
#else
    (void)wg;
    float* accs = new float[(size_t)Md * Nd]();
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        for (int row = 0; row < (g1 - g0) * rows_per_tap; ++row)
            for (int c = 0; c < F; ++c) wbuf[row * ldw + c] = wsrc[(g0 * rows_per_tap + row) * F + c];
        for (int m = 0; m < Md; ++m)
            for (int n = 0; n < Nd; ++n) {
                float acc = accs[(size_t)m * Nd + n];
                for (int ko = g0; ko < g1; ++ko)
                    for (int ki = 0; ki < Ki; ++ki)
                        acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko - g0, ki & ~3), acc);
                accs[(size_t)m * Nd + n] = acc;
            }
    }
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < Nd; ++n) fc.put(m, n, accs[(size_t)m * Nd + n]);
    delete[] accs;
#endif
}

// ---- MODE 3 (round 5): the conv products of the long protein CNNs, F = 32 filters ---------------------------------------------
// Measured in round 5's first session (profiles/r5_train_gfp_*): the staged form (MODE 2) runs conv3 forward at 0.36 and its
// backward at 0.30 of the f32-MFMA rate -- one A and one B ds_read (the B one 2-way conflicted: rows F + 4 apart) and ~10 address
// instructions per MFMA, the staging copy's L2 round trip exposed between two barriers per tap group, and the weight gradient
// (78 tiles x 59 k-steps) fetching both operands per MFMA through the heaviest index functors.  Three changes, same bits:
//   * fxt_conv32_staged: a wave owns a 16-row M tile and BOTH 16-column N tiles (F = 32): one A fetch feeds two MFMAs.  The tap
//     group's kernels are kept in LDS with 32-float rows, rotated so that the fetch is conflict-free WITHOUT padding -- forward:
//     element (c, n) at column (n + 16 c) mod 32, transposed read of the input gradient: element (c, o) at column (o + 2 c) mod 32
//     (a product stages its own copy, so each picks its rotation) -- seven taps per group instead of six in the same bytes; the
//     next group's rows are fetched from L2 into registers BEFORE the current group's MFMAs and stored behind them;
//   * fxt_conv32_wgrad: dW[j][c][o] = sum_t x[t + j - pl][c] dz[t][o].  Taps j and j + 4 read the same x k-step blocks one k-step
//     apart, so a wave owns taps {res, res + 4, ...} x 16 channels x 16 out channels (16 jobs = 4 residues x 2 x 2): per k-step ONE
//     new x block and ONE dz block from LDS feed up to five MFMAs out of a register window.  The bias row rides on the residue-3
//     waves (four taps of conv3's nineteen).
// Every output element still sums the same products in the same order through the same instruction: the SAME BITS as fxt_gemm.
// The tap group in the staging buffer, FRAGMENT order: a tap is 2 halves x 32 columns x 16 floats; the sixteen floats of (half h, column n)
// are [kq'][u] with kq' = (kq + 2 ((n >> 3) & 1)) mod 4 (the swizzle keeps the sixteen lanes of a ds_read_b128 lane group on distinct
// 16-byte bank groups) and hold the contraction elements 16 h + 4 u + kq: forward B((j, c), n) = W[j][c][n] with c the contraction,
// input gradient B((j, o), c) = W[j][c][o] with o the contraction and c the column.  A lane's operand for a half-tap is one 16-byte read
// at a per-lane offset plus a wave-uniform one.
template <class P, class P4>
struct FxtConvW4 {
    P w;
    FXT_HD int prep(int n, int kq) const { return n * 16 + 4 * ((kq + 2 * ((n >> 3) & 1)) & 3); }
    FXT_HD float at(int st, int j, int k0) const { return w[st + (j * 2 + (k0 >> 4)) * 512 + ((k0 >> 2) & 3)]; }       // (scalar form: host build)
    FXT_HD auto at4(int st, int j, int h) const { return *(P4)(w + st + (j * 2 + h) * 512); }
};
// where element (row = jrel 32 + c, column e) of a kernel's tap group goes.  ROT 0: forward (contraction = row's channel c, column n = e);
// 1: input gradient (contraction = e, column = c)
template <int ROT>
FXT_HD int fxt_w4_off(int row, int e) {
    const int jrel = row >> 5, c = row & 31;
    const int k = ROT ? e : c, n = ROT ? c : e;              // contraction element, column
    return ((jrel * 2 + (k >> 4)) * 32 + n) * 16 + 4 * (((k & 3) + 2 * ((n >> 3) & 1)) & 3) + ((k >> 2) & 3);
}
FXT_HD bool fxt_conv32_ok(int Md, int F, int nw) { return F == 32 && ((Md + 15) >> 4) <= nw; }
// A tap group on its way from L2 to the staging buffer: PF 16-byte pieces per thread in registers.  Carried ACROSS products and phase
// barriers -- conv2's group is fetched while conv1 runs, conv3's first group behind conv2's MFMAs, conv3's input-gradient group while
// the max-pool backward runs, conv2's behind conv3's input gradient -- so that no product starts by waiting for L2, and (backward) no
// weight fetch is issued behind a phase's gradient-partial stores: vmcnt retires in order, a load issued after 78 KiB of partial stores
// waits for all of them (round 5: conv2's backward phase took 30 us for ~10 us of work in every form; this was why).
#define FXT_TAP_PF 2
// A piece = the sixteen bytes one lane of the product will read as ONE operand: piece p of a group is (tap p >> 8, half (p >> 7) & 1,
// column (p >> 2) & 31, lane group kq = p & 3) and holds the contraction elements 16 half + 4 u + kq, u = 0 .. 3 -- four dword loads
// (64-byte runs across the lanes) and ONE ds_write_b128 into the fragment order of FxtConvW4, conflict-free.  (Fetched as 16-byte
// row pieces and scattered by four ds_write_b32 the stores hit the banks 8-way: ~0.9 us per group between two barriers, round 5.)
template <int WAS>
struct FxtTapRegs {
#if FXT_DEVICE
    fxt_f4 pre[FXT_TAP_PF];
#endif
    int taps;                                              // taps held (0 = nothing)
    // ROT 0: forward (contraction = the kernel's input channel, column = output channel); 1: input gradient (the other way round)
    template <int ROT>
    FXT_HD void fetch(const FxtWg& wg, typename FxtMem<WAS>::CF wsrc, int g0, int g1) {      // taps [g0, g1) of a 32 x 32-per-tap kernel
        taps = g1 - g0;
#if FXT_DEVICE
        const int pieces = taps * 256;
#pragma unroll
        for (int q = 0; q < FXT_TAP_PF; ++q) {
            const int p = wg.tid + q * wg.nthr;
            if (p < pieces) {
                const int jr = p >> 8, half = (p >> 7) & 1, col = (p >> 2) & 31, kq = p & 3;
                typename FxtMem<WAS>::CF src = wsrc + (g0 + jr) * 1024 + (ROT ? col * 32 + half * 16 + kq : (half * 16 + kq) * 32 + col);
#pragma unroll
                for (int u = 0; u < 4; ++u) pre[q][u] = src[(ROT ? 4 : 128) * u];
            }
        }
#else
        (void)wg; (void)wsrc; (void)g0;
#endif
    }
    template <int WSAS, class WB>
    FXT_HD void store(const FxtWg& wg, WB wbuf) {
#if FXT_DEVICE
        const int pieces = taps * 256;
#pragma unroll
        for (int q = 0; q < FXT_TAP_PF; ++q) {
            const int p = wg.tid + q * wg.nthr;
            if (p < pieces) {
                const int blk = p >> 2, col = blk & 31, kq = p & 3;         // blk = (tap 2 + half) 32 + column
                *(typename FxtMem<WSAS>::F4)(wbuf + blk * 16 + 4 * ((kq + 2 * ((col >> 3) & 1)) & 3)) = pre[q];
            }
        }
#else
        (void)wg; (void)wbuf;
#endif
        taps = 0;
    }
};
// taps per group: what the host sized the buffer for, and what FXT_TAP_PF pieces per thread carry
FXT_HD int fxt_conv32_group(int stage_taps, int nthr) {
    const int gmax = FXT_TAP_PF * nthr / 256;
    int G = stage_taps < gmax ? stage_taps : gmax;
    return G < 1 ? 1 : G;
}
// ROT: 0 = forward rotation (16 c), 1 = input-gradient rotation (2 c).  `G` = fxt_conv32_group taps per group.
// tap: the register carrier.  pre_loaded: it holds taps [0, min(G, Ko)) already.  in_wbuf: those taps are in wbuf already (stored and
// published by an earlier barrier).  next_src / next_Ko: the NEXT product's kernel -- its first group is fetched into `tap` behind this
// product's last group of MFMAs and left there.
struct FxtNoDbg { FXT_HD void operator()(int) const {} };
template <int WSAS, int WAS, int ROT, class FA, class FB, class FC, class DBG = FxtNoDbg>
FXT_HD void fxt_conv32_staged(const FxtWg& wg, int Md, int Ko, const FA& fa, const FB& fb, const FC& fc,
                              typename FxtMem<WAS>::CF wsrc, typename FxtMem<WSAS>::F wbuf, int G, FxtTapRegs<WAS>& tap,
                              bool pre_loaded = false, bool in_wbuf = false, typename FxtMem<WAS>::CF next_src = nullptr, int next_Ko = 0,
                              const DBG& dbg = DBG()) {
    if (G < 1) G = 1;
#if FXT_DEVICE
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int U = 8;
    const int lane = wg.tid & 63, wave = fxt_wave(wg);
    const int i = lane & 15, kq = lane >> 4;
    const int tm = (Md + 15) >> 4;
    const bool have = wave < tm;                           // (wave-uniform; fxt_conv32_ok: every M tile has its wave)
    const int m = wave * 16 + i;
    const auto sa = fa.prep((have && m < Md) ? m : 0, kq);
    const auto sb0 = fb.prep(i, kq), sb1 = fb.prep(i + 16, kq);
    f4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const auto e0 = fc.pre(i), e1 = fc.pre(i + 16);        // what the epilogue needs per column (a bias from global memory): fetched now, not in a dependent chain at the end
    // half a tap's operands: four k-steps of A and of both B tiles.  Half-tap h + 1's are fetched BEFORE half-tap h's eight MFMAs (two
    // register sets of twelve, the loop unrolled by two): an LDS round trip hides behind ~256 cycles of matrix work instead of
    // preceding it.  (Whole taps in flight -- 2 x 24 registers -- spilled at the 128 registers sixteen waves leave a thread.)
    constexpr int UH = U / 2;
    struct Ops { f4_t a, b0, b1; };                        // (one 16-byte LDS read each: fxt_xi<2>, FxtConvW4)
    auto load = [&](Ops& o, int h, int g0) {               // half-tap h of the group: tap g0 + h / 2, k-steps 4 (h & 1) ...
        const int ko = g0 + (h >> 1);
        o.a = fa.at4(sa, ko, h & 1); o.b0 = fb.at4(sb0, ko - g0, h & 1); o.b1 = fb.at4(sb1, ko - g0, h & 1);
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[u], o.b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[u], o.b1[u], acc1, 0, 0, 0);
        }
    };
    if (!pre_loaded && !in_wbuf) tap.template fetch<ROT>(wg, wsrc, 0, G < Ko ? G : Ko);
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        if (!(in_wbuf && g0 == 0)) {
            fxt_sync_ws<WSAS>();                           // everybody is through with the previous group's taps (or the previous phase)
            tap.template store<WSAS>(wg, wbuf);
        }
        if (g1 < Ko) tap.template fetch<ROT>(wg, wsrc, g1, g1 + G < Ko ? g1 + G : Ko);       // in flight behind this group's MFMAs
        else if (next_src) tap.template fetch<ROT>(wg, next_src, 0, G < next_Ko ? G : next_Ko);
        fxt_sync_ws<WSAS>();
        dbg(2 * (g0 / G));                                 // (profiling aid: this wave's clock before / after a group's MFMAs)
        if (!have) continue;
        Ops x, y;
        const int H = 2 * (g1 - g0);                       // (even)
        load(x, 0, g0);
        for (int h = 0; h < H; h += 2) {
            load(y, h + 1, g0);
            FXT_SCHED_FENCE();                             // (the fetches stay IN FRONT of the MFMAs they hide behind: left alone, the scheduler sinks each one to just before its use)
            mma(x);
            load(x, h + 2 < H ? h + 2 : h, g0);            // (past the group's end: the same half-tap again, unused -- a straight-line body lets the waits be counted exactly)
            FXT_SCHED_FENCE();
            mma(y);
        }
        dbg(2 * (g0 / G) + 1);
    }
    if (have) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = wave * 16 + 4 * kq + r;
            if (mr < Md) { fc.put(mr, i, acc0[r], e0); fc.put(mr, i + 16, acc1[r], e1); }
        }
    }
#else
    (void)wg; (void)tap; (void)pre_loaded; (void)in_wbuf; (void)next_src; (void)next_Ko;     // (the host build stages every group itself: the same values)
    float* accs = new float[(size_t)Md * 32]();
    for (int g0 = 0; g0 < Ko; g0 += G) {
        const int g1 = g0 + G < Ko ? g0 + G : Ko;
        for (int row = 0; row < (g1 - g0) * 32; ++row)
            for (int c = 0; c < 32; ++c) wbuf[fxt_w4_off<ROT>(row, c)] = wsrc[(g0 * 32 + row) * 32 + c];
        for (int m = 0; m < Md; ++m)
            for (int n = 0; n < 32; ++n) {
                float acc = accs[(size_t)m * 32 + n];
                for (int ko = g0; ko < g1; ++ko)
                    for (int ki = 0; ki < 32; ++ki)
                        acc = fmaf(fa.at(fa.prep(m, ki & 3), ko, ki & ~3), fb.at(fb.prep(n, ki & 3), ko - g0, ki & ~3), acc);
                accs[(size_t)m * 32 + n] = acc;
            }
    }
    for (int m = 0; m < Md; ++m)
        for (int n = 0; n < 32; ++n) fc.put(m, n, accs[(size_t)m * 32 + n], fc.pre(n));
    delete[] accs;
#endif
}

// Conv weight gradient of a 32 -> 32 channel layer over ROTATED rows (see above): x, dz position-major arrays of R x L1 rows,
// fc.put(row (j 32 + c, or Kt 32 for the bias), column o, value).  Kt <= 20 taps.  `mid`: called by every wave ONCE, after its (first)
// job's loop and before any of its gradient stores (the caller commits a prefetched tap group there: a barrier inside).
// KT > 0: the tap count as a compile-time constant (canonical instantiations): a job's taps are then a constant per residue, its
// MFMAs unconditional, and the register window turns over by renaming inside blocks of five k-steps instead of by moves.
#define FXT_WG32_MAXT 5
struct FxtNoHook { FXT_HD void operator()() const {} };
#if FXT_DEVICE
typedef float fxt_acc4 __attribute__((ext_vector_type(4)));
// one job's loops: taps res, res + 4, ... (NT of them; NT < 0: a run-time count `ntr`), channel tile at c, out-channel tile at o
template <int NT, bool BIAS, class P>
FXT_HD void fxt_wg32_job(int R, int L1, int res, int pl, int c, int o, int kq, int ntr, bool biasr, P x, P dz, P zero,
                         fxt_acc4 (&acc)[FXT_WG32_MAXT], fxt_acc4& accb) {
    constexpr int MT = FXT_WG32_MAXT;
    const int S = (L1 + 3) >> 2, Sfull = L1 >> 2;          // k-steps of a row; those whose four positions all lie inside it
    for (int rho = 0; rho < R; ++rho) {
        const int base = rho * L1;
        // x block b: position 4 b + kq + res - pl of the row, channel c (zero outside the row).  Fetch and mask are separate steps:
        // the mask is applied where the value is USED, one k-step later, so nothing waits on LDS between a fetch and the MFMAs of
        // the k-step it is issued in.
        auto Xok = [&](int b) { const int pp = 4 * b + kq + res - pl; return pp >= 0 && pp < L1; };
        // (positions outside the row are read from `zero`, a row of zeros in LDS: no select behind the fetch)
        auto Xraw = [&](int b, bool ok) {
            const int pp = 4 * b + kq + res - pl;
            return *(ok ? x + fxt_xi<2>(base + pp, c, 32) : zero);
        };
        auto Braw = [&](int s, bool ok) {
            const int t = 4 * s + kq;
            return *(ok ? dz + fxt_xi<2>(base + t, o, 32) : zero);
        };
        float win[MT];                                     // win[(s + q) % MT] = x block s + q inside the blocks of MT k-steps below
#pragma unroll
        for (int q = 0; q < MT - 1; ++q) win[q] = Xraw(q, Xok(q));
        // two k-steps of operands in flight: (xr, br) for the step about to run, (xr2, br2) for the one after -- one step ahead covers
        // an LDS round trip only behind five MFMAs; conv2's jobs issue one or two per k-step (round 5: 345 cycles per k-step there)
        bool bok = kq < L1, bok2 = 4 + kq < L1;
        float xr = Xraw(MT - 1, Xok(MT - 1)), br = Braw(0, bok);
        float xr2 = Xraw(MT, Xok(MT)), br2 = Braw(1, bok2);
        int s = 0;
        if constexpr (NT >= 0) {
            for (; s + MT <= Sfull; s += MT) {             // MT whole k-steps: the window's slots are compile-time constants
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    win[(u + MT - 1) % MT] = xr;
                    const float b = br;
                    xr = xr2; br = br2; bok = bok2;
                    {   const int sn = s + u + 2;          // the fetches of the k-step after next (masked past the row's end)
                        bok2 = 4 * sn + kq < L1;
                        xr2 = Xraw(sn + MT - 1, Xok(sn + MT - 1)); br2 = Braw(sn, bok2); }
                    FXT_SCHED_FENCE();
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[(u + q) % MT], b, acc[q], 0, 0, 0);
                    if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, b, accb, 0, 0, 0);
                }
            }
        }
        const int nt = NT >= 0 ? NT : ntr;
        const bool bias = NT >= 0 ? BIAS : biasr;
        for (; s < S; ++s) {                               // the rest of the row (run-time tap counts: all of it): the window moves by copies
            win[MT - 1] = xr;
            const float b = br;
            const bool tok = bok;                          // 4 s + kq < L1: false only in a row's last, partial k-step -- both operands zero there, as fxt_gemm masks them
            xr = xr2; br = br2; bok = bok2;
            bok2 = 4 * (s + 2) + kq < L1;
            xr2 = Xraw(s + MT + 1, Xok(s + MT + 1)); br2 = Braw(s + 2, bok2);
#pragma unroll
            for (int q = 0; q < MT; ++q)
                if (q < nt) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(tok ? win[q] : 0.f, b, acc[q], 0, 0, 0);
            if (bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(tok ? 1.f : 0.f, b, accb, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MT - 1; ++q) win[q] = win[q + 1];
        }
    }
}
#endif
template <int KT = 0, class P, class FC, class HOOK = FxtNoHook>
FXT_HD void fxt_conv32_wgrad(const FxtWg& wg, int R, int L1, int Kt, int pl, P x, P dz, P zero, const FC& fc, const HOOK& mid = HOOK()) {
#if FXT_DEVICE
    constexpr int MT = FXT_WG32_MAXT;
    const int lane = wg.tid & 63, nw = wg.nthr >> 6;
    const int i = lane & 15, kq = lane >> 4;
    bool hooked = false;
    for (int job = fxt_wave(wg); job < 16; job += nw) {    // (sixteen waves: one job each)
        const int res = job >> 2, ct = (job >> 1) & 1, ot = job & 1;
        const int NT = Kt > res ? (Kt - res + 3) >> 2 : 0; // taps res, res + 4, ... of this job (wave-uniform)
        const bool bias = res == 3 && ct == 0;
        const int c = ct * 16 + i, o = ot * 16 + i;
        fxt_acc4 acc[MT], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < MT; ++q) acc[q] = fxt_acc4{0.f, 0.f, 0.f, 0.f};
        if constexpr (KT > 0) {
            constexpr int N0 = (KT + 3) >> 2, N1 = KT > 1 ? (KT + 2) >> 2 : 0, N2 = KT > 2 ? (KT + 1) >> 2 : 0, N3 = KT > 3 ? KT >> 2 : 0;
            if (res == 0) fxt_wg32_job<N0, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (res == 1) fxt_wg32_job<N1, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (res == 2) fxt_wg32_job<N2, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else if (bias) fxt_wg32_job<N3, true>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
            else fxt_wg32_job<N3, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
        } else {
            fxt_wg32_job<-1, false>(R, L1, res, pl, c, o, kq, NT, bias, x, dz, zero, acc, accb);
        }
        if (!hooked) { mid(); hooked = true; }
#pragma unroll
        for (int q = 0; q < MT; ++q)
            if (q < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fc.put((res + 4 * q) * 32 + ct * 16 + 4 * kq + r, o, acc[q][r]);
            }
        if (bias && kq == 0) fc.put(Kt * 32, o, accb[0]);
    }
    if (!hooked) mid();                                    // (more than sixteen waves: the rest still meets the hook's barrier)
#else
    (void)wg; (void)zero;
    mid();
    for (int mrow = 0; mrow <= Kt * 32; ++mrow)
        for (int o = 0; o < 32; ++o) {
            const int j = mrow >> 5, c = mrow & 31;
            float acc = 0.f;
            for (int rho = 0; rho < R; ++rho)
                for (int t = 0; t < L1; ++t) {
                    const int pp = t + j - pl;
                    const float a = mrow == Kt * 32 ? 1.f : ((pp >= 0 && pp < L1) ? x[fxt_xi<2>(rho * L1 + pp, c, 32)] : 0.f);
                    acc = fmaf(a, dz[fxt_xi<2>(rho * L1 + t, o, 32)], acc);
                }
            fc.put(mrow, o, acc);
        }
#endif
}

// ---- operand functors ------------------------------------------------------------------------------------------
// A k-step covers contraction indices ki = k0 + kq, kq = lane >> 4 in 0..3, k0 wave-uniform.  Every functor splits its
// address into a per-lane part (`prep`, once per tile: row / column decomposition, the kq term) and a wave-uniform part
// built from (ko, k0) in `at` -- scalar arithmetic on the GPU -- so that an operand fetch costs one or two vector
// instructions.  (The first build recomputed the whole index per element: ~12 VALU instructions per operand, and with
// four waves per SIMD the address arithmetic, not the memory, set the step time.)
template <class P>
struct FxtRowMajorA {          // A(m, 0, ki) = p[m * ld + ki]
    P p; int ld;
    FXT_HD int prep(int m, int kq) const { return m * ld + kq; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0]; }
};
template <class P>
struct FxtRowMajorB {          // B(0, ki, n) = p[ki * ld + n]
    P p; int ld;
    FXT_HD int prep(int n, int kq) const { return kq * ld + n; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0 * ld]; }
};
template <class P>
struct FxtTransB {             // B(0, ki, n) = p[n * ld + ki]      (W^T for the input gradients)
    P p; int ld;
    FXT_HD int prep(int n, int kq) const { return n * ld + kq; }
    FXT_HD float at(int st, int, int k0) const { return p[st + k0]; }
};
// conv forward: rows m = (r, t), contraction (tap j, channel c): A = x[r][t + j - pl][c] inside the sequence, else 0
template <class P>
struct FxtConvA {               // (C = x's row stride)
    P x; int Lx, C, pl; FxtDiv dL;
    struct St { int base, tp; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{(m - pl) * C + kq, t - pl}; }
    FXT_HD float at(St s, int j, int k0) const {
        const int p = s.tp + j;
        const bool ok = p >= 0 && p < Lx;
        const float v = x[ok ? s.base + j * C + k0 : 0];             // (clamped index + select: no branch around the load)
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvW {              // B((j, c), n) = w[(j * C + c) * F + n]      (F = the kernel's row stride)
    P w; int C, F;
    FXT_HD int prep(int n, int kq) const { return kq * F + n; }
    FXT_HD float at(int st, int j, int k0) const { return w[st + (j * C + k0) * F]; }
};
// conv input gradient: rows m = (r, s), contraction (tap j, out channel o): A = dz[r][s - j + pl][o], B = w[j][n][o]
template <class P>
struct FxtConvGradA {           // (F = dz's row stride)
    P dz; int Lx, F, pl; FxtDiv dL;
    struct St { int base, sp; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{(m + pl) * F + kq, s + pl}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        const bool ok = p >= 0 && p < Lx;
        const float v = dz[ok ? st.base - j * F + k0 : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvGradW {          // B((j, o), n = c) = w[(j * C + c) * F + o]  (F = the kernel's row stride)
    P w; int C, F;
    FXT_HD int prep(int n, int kq) const { return n * F + kq; }
    FXT_HD float at(int st, int j, int k0) const { return w[st + j * C * F + k0]; }
};
// conv weight gradient: rows m = (tap j, channel c) plus ONE extra row for the bias; contraction (row r, position t)
template <class P>
struct FxtConvWGradA {
    P x; int Lx, C, ld, pl, rows; FxtDiv dC; // rows = taps * C (row `rows` is the bias row: all ones); ld = x's row stride
    struct St { int off, tp; };              // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0};
        const int j = fxt_quot(m, dC), c = m - j * C;
        return St{(j - pl + kq) * ld + c + (1 << 30), j - pl + kq};     // (+2^30: keeps `off` non-negative for taps left of the sequence)
    }
    FXT_HD float at(St s, int r, int k0) const {
        const int p = s.tp + k0;
        const bool ok = s.off >= 0 && p >= 0 && p < Lx;
        const float v = x[ok ? s.off - (1 << 30) + (r * Lx + k0) * ld : 0];
        return s.off < 0 ? 1.f : (ok ? v : 0.f);
    }
};
// conv1 / first dense layer: x is the one-hot of the codes.  Rows m = (j, c) = m / A, m % A plus the bias row.
// conv = 1: contraction (ko = row r, ki = position t), element [code[r][t + j] == c];
// conv = 0: contraction (ko = 0, ki = row r),          element [code[r][j] == c]   (j = the position of input unit m)
template <class P>
struct FxtOneHotWGradA {
    P codes; int L, A, rows, conv; FxtDiv dA;
    struct St { int off, c; };               // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0};
        const int j = fxt_quot(m, dA), c = m - j * A;
        return St{conv ? j + kq : kq * L + j, c};
    }
    FXT_HD float at(St s, int ko, int k0) const {
        const int code = codes[s.off < 0 ? 0 : s.off + (conv ? ko * L + k0 : k0 * L)];
        return (s.off < 0 || code == s.c) ? 1.f : 0.f;
    }
};
template <class P>
struct FxtPosMajorB {          // B((r, t), n) = p[(r * Lx + t) * F + n]     (F = p's row stride)
    P p; int Lx, F;
    FXT_HD int prep(int n, int kq) const { return kq * F + n; }
    FXT_HD float at(int st, int r, int k0) const { return p[st + (r * Lx + k0) * F]; }
};
// dense weight gradient: rows m = input unit k plus the bias row; contraction over the slice's rows r
template <class P>
struct FxtDenseWGradA {
    P in; int Kd, ld;          // ld = in's row stride
    FXT_HD int prep(int m, int kq) const { return m >= Kd ? -1 : kq * ld + m; }
    FXT_HD float at(int st, int, int k0) const { const float v = in[st < 0 ? 0 : st + k0 * ld]; return st < 0 ? 1.f : v; }
};

// ---- the same operands over ROTATED rows (fxt_xi<true>; the row stride equals the channel count, a power of two) ----
template <class P>
struct FxtConvASwz {            // element (row m - pl + j, channel k0 + kq)
    P x; int Lx, C, pl; FxtDiv dL;
    struct St { int base, tp, rot; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{(m - pl) * C, t - pl, kq + 2 * (m - pl)}; }
    FXT_HD float at(St s, int j, int k0) const {
        const int p = s.tp + j;
        const bool ok = p >= 0 && p < Lx;
        const float v = x[ok ? s.base + j * C + ((s.rot + 2 * j + k0) & (C - 1)) : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvGradASwz {        // element (row m + pl - j, channel k0 + kq)
    P dz; int Lx, F, pl; FxtDiv dL;
    struct St { int base, sp, rot; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{(m + pl) * F, s + pl, kq + 2 * (m + pl)}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        const bool ok = p >= 0 && p < Lx;
        const float v = dz[ok ? st.base - j * F + ((st.rot - 2 * j + k0) & (F - 1)) : 0];
        return ok ? v : 0.f;
    }
};
template <class P>
struct FxtConvWGradASwz {       // element (row r Lx + k0 + j - pl + kq, channel c)
    P x; int Lx, C, ld, pl, rows; FxtDiv dC;
    struct St { int off, tp, rot; };         // off < 0: bias row
    FXT_HD St prep(int m, int kq) const {
        if (m >= rows) return St{-1, 0, 0};
        const int j = fxt_quot(m, dC), c = m - j * C;
        return St{(j - pl + kq) * ld + (1 << 30), j - pl + kq, c + 2 * (j - pl + kq)};
    }
    FXT_HD float at(St s, int r, int k0) const {
        const int p = s.tp + k0;
        const bool ok = s.off >= 0 && p >= 0 && p < Lx;
        const int ru = r * Lx + k0;
        const float v = x[ok ? s.off - (1 << 30) + ru * ld + ((s.rot + 2 * ru) & (ld - 1)) : 0];
        return s.off < 0 ? 1.f : (ok ? v : 0.f);
    }
};
template <class P>
struct FxtPosMajorBSwz {        // element (row r Lx + k0 + kq, channel n)
    P p; int Lx, F;
    struct St { int off, rot; };
    FXT_HD St prep(int n, int kq) const { return St{kq * F, n + 2 * kq}; }
    FXT_HD float at(St st, int r, int k0) const { const int ru = r * Lx + k0; return p[st.off + ru * F + ((st.rot + 2 * ru) & (F - 1))]; }
};

// MODE 3: the conv A operands over FRAGMENT rows (fxt_xi<2>), positions outside the row sent to a row of zeros in LDS (`zoff`: its index relative to the
// array) instead of a select on the fetched value -- the fetch then has no consumer but its MFMA, so it can be issued a half-tap ahead
// (a select right behind the fetch made the compiler wait for LDS there), and four instructions per half-tap go away.  0.0 either way.
template <class P, class P4>
struct FxtConvAZ {              // conv forward over fragment rows: element (row m - pl + j, channel k0 + kq)
    P x; int Lx, pl, zoff; FxtDiv dL;
    struct St { int row, tp, kq; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), t = m - r * Lx; return St{m - pl, t - pl, kq}; }
    FXT_HD float at(St s, int j, int k0) const {             // (scalar form: host build)
        const int p = s.tp + j;
        return (p >= 0 && p < Lx) ? x[fxt_xi<2>(s.row + j, k0 + s.kq, 32)] : 0.f;
    }
    FXT_HD auto at4(St s, int j, int h) const {              // channels 16 h + 4 u + kq, u = 0 .. 3: one 16-byte read
        const int p = s.tp + j, row = s.row + j;
        const bool ok = p >= 0 && p < Lx;
        return *(P4)(x + (ok ? row * 32 + 4 * ((4 * h + s.kq + (row & 6)) & 7) : zoff));
    }
};
template <class P, class P4>
struct FxtConvGradAZ {          // conv input gradient over fragment rows: element (row m + pl - j, channel k0 + kq)
    P dz; int Lx, pl, zoff; FxtDiv dL;
    struct St { int row, sp, kq; };
    FXT_HD St prep(int m, int kq) const { const int r = fxt_quot(m, dL), s = m - r * Lx; return St{m + pl, s + pl, kq}; }
    FXT_HD float at(St st, int j, int k0) const {
        const int p = st.sp - j;
        return (p >= 0 && p < Lx) ? dz[fxt_xi<2>(st.row - j, k0 + st.kq, 32)] : 0.f;
    }
    FXT_HD auto at4(St st, int j, int h) const {
        const int p = st.sp - j, row = st.row - j;
        const bool ok = p >= 0 && p < Lx;
        return *(P4)(dz + (ok ? row * 32 + 4 * ((4 * h + st.kq + (row & 6)) & 7) : zoff));
    }
};
// B((r, t), n) over any layout (conv1's weight gradient reads the fragment rows through the shape-agnostic product)
template <class P, int LAY>
struct FxtPosMajorBL {
    P p; int Lx, F;
    struct St { int n, kq; };
    FXT_HD St prep(int n, int kq) const { return St{n, kq}; }
    FXT_HD float at(St st, int r, int k0) const { return p[fxt_xi<LAY>(r * Lx + k0 + st.kq, st.n, F)]; }
};

// Compile-time shape of a CANONICAL network (round 4).  The step is written for any shape the constructors accept: every
// contraction chooses among three k-step walks at run time, masks its overhangs, and builds its addresses from run-time
// dimensions -- 100 KiB of code per placement, executed once per launch, i.e. streamed through the 64 KiB instruction cache
// every step, and ~30 non-MFMA instructions per MFMA (profiles/r3_train_pmc.md).  For the shapes the explorers' surrogates are
// actually built with (CNN(32, 100, kernel 5) on 4 letters, MLP(100), GlobalEpistasis(100)) the same
// source is instantiated with the dimensions as constants: dead walks and masks fold away, offsets become immediates.  Same
// arithmetic in the same order: the SAME BITS as the generic instantiation (GPU test).
struct FxtDimsAny { static constexpr bool fixed = false; static constexpr int kind = 0, A = 0, F = 0, H = 0, K = 0, R = 0, L = 0; };
template <int KIND, int A_, int F_, int H_, int K_, int R_, int L_ = 0>     // L_ = 0 / R_ = 0: the sequence length / the rows per slice stay run-time values
struct FxtDims { static constexpr bool fixed = true; static constexpr int kind = KIND, A = A_, F = F_, H = H_, K = K_, R = R_, L = L_; };

// ---------------------------------------------------------------------------------------------------------------
// Forward + backward of one slice.  `slice` rows [slice * R, slice * R + R) of the mini-batch `step`.
// `ws`: the slice's workspace -- the workgroup's LDS on the device when it fits (activations are written by one phase
// and read by the next: an LDS round trip instead of an L2 one; WSAS = 3), else its row of the global arena (WSAS = 1).
// `W`: the member's weights, staged in LDS by the caller when they fit next to the workspace (WAS = 3), else j.w.
template <int WSAS, int WAS, class D = FxtDimsAny, int MODE = 0>
FXT_HD void fxt_forward_backward(const FxtJob& j, const FxtWg& wg, int step, int slice, const uint8_t* ascii,
                                 const uint8_t* lut, const float* labels, typename FxtMem<WSAS>::F ws,
                                 typename FxtMem<WAS>::CF W, typename FxtMem<WSAS>::F split = nullptr) {
    typedef typename FxtMem<WSAS>::F WsF;
    typedef typename FxtMem<WSAS>::CF WsCF;
    typedef typename FxtMem<WSAS>::I WsI;
    typedef typename FxtMem<WSAS>::CI WsCI;
    typedef typename FxtMem<WAS>::CF WCF;
    constexpr bool SWZ = MODE != 0;
    constexpr int LAY = MODE == 3 ? 2 : (MODE != 0 ? 1 : 0);      // how the position-major arrays are indexed (fxt_xi)
    typedef typename FxtMem<WSAS>::CF4 WsCF4;
    typedef typename FxtPick<SWZ, FxtConvA<WsCF>, FxtConvASwz<WsCF>>::T ConvA;
    typedef typename FxtPick<SWZ, FxtConvGradA<WsCF>, FxtConvGradASwz<WsCF>>::T ConvGradA;
    typedef typename FxtPick<SWZ, FxtConvWGradA<WsCF>, FxtConvWGradASwz<WsCF>>::T ConvWGradA;
    typedef typename FxtPick<SWZ, FxtPosMajorB<WsCF>, FxtPosMajorBSwz<WsCF>>::T PosMajorB;
    // (a canonical instantiation rebuilds the description from its constants -- only the sequence length is a run-time value --
    //  so that everything derived from it below is a constant too)
    FxtNet n_ = D::fixed ? fxt_net(D::kind, D::L > 0 ? D::L : j.net.L, D::A, D::F, D::H, D::K) : j.net;
    if (D::fixed && SWZ) n_.ldx = n_.F;        // (rotated rows are F floats apart)
    const FxtNet n = n_;
    const int R = (D::fixed && D::R > 0) ? D::R : j.R, L = n.L, A = n.A, F = n.F;      // (R_ = 0: the rows per slice stay a run-time value too)
    const FxtWs w = fxt_ws(n, R, MODE >= 2);
    // MODE 2 / 3: the conv kernels' staging buffer behind the workspace, j.split_off taps at a time (the host sized it: a tap is
    // F rows of fxt_ld_w(F) floats in MODE 2, 32 rotated rows of 32 floats in MODE 3)
    [[maybe_unused]] WsF wbuf = ws + w.total;
    [[maybe_unused]] const int stage_taps = j.split_off;
    [[maybe_unused]] const int tap_floats = MODE == 3 ? F * F : F * fxt_ld_w(F);
    [[maybe_unused]] FxtTapRegs<WAS> tap;                  // MODE 3: the tap group in flight (see FxtTapRegs)
    tap.taps = 0;
    [[maybe_unused]] const int G3 = fxt_conv32_group(stage_taps, wg.nthr);
    const FxtLay y = fxt_lay(n, WAS == 3);      // (the LDS image of the weights has padded conv-kernel rows)
    const int ldF = w.ldF, ldw = y.ldw;
    WsI codes = (WsI)(ws + w.codes);
    float* part = j.partial + (long long)slice * fxt_pstride(j);
    const int32_t* order = j.order + (long long)step * j.batch;
    const int slot0 = slice * R;
    const int nwv = wg.nthr >> 6;
    const bool can_split = split != nullptr;
    const int sidx = step % j.steps_per_epoch;
    const int nvalid = (j.n - sidx * j.batch) < j.batch ? (j.n - sidx * j.batch) : j.batch;
    FXT_STAMP(0);
    const float keep_scale = 1.f / (1.f - FXT_DROPOUT);
    const FxtDiv dL1 = fxt_div(n.kind == 0 ? n.L1 : 1), dF = fxt_div(n.kind == 0 ? F : 1), dA = fxt_div(A);

    if constexpr (MODE == 3) {
        if (n.kind == 0) {
            tap.template fetch<0>(wg, W + y.cw[1], 0, G3 < n.K ? G3 : n.K);      // conv2's first tap group: two phases ahead
            FXT_FOR(i, w.ldF, wg) ws[w.zero + i] = 0.f;              // the row of zeros (FxtConvAZ)
        }
    }
    // ---- the slice's rows as alphabet indices (padding slots read row 0: their gradient is zeroed at the loss)
    FXT_FOR(i, R * L, wg) {
        const int r = i / L, l = i - r * L;
        const int slot = slot0 + r;
        const int row = (slot < j.batch && order[slot] >= 0) ? order[slot] : 0;
        codes[i] = lut[ascii[(long long)row * L + l]];
    }
    FXT_FOR(r, R, wg) {                     // (the label fetch is a dependent global load too: issued here, used after the forward)
        const int slot = slot0 + r;
        const bool valid = slot < j.batch && order[slot] >= 0;
        ws[w.ylab + r] = labels[valid ? order[slot] : 0];
        ws[w.yvalid + r] = valid ? 1.f : 0.f;
    }
    // MODE 2: conv1's kernel and bias (K A rows of F floats + F: contiguous in Keras order) into the staging buffer as well, under the
    // same barrier -- conv1 is K gathered kernel rows per output, from L2 otherwise (~7 dependent round trips per thread at one row per slice)
    [[maybe_unused]] bool conv1_staged = false;
    if constexpr (MODE >= 2) {
        if (n.kind == 0 && (n.K * A + 1) * F <= stage_taps * tap_floats) {
            conv1_staged = true;
            FXT_FOR(i, (n.K * A + 1) * F, wg) wbuf[i] = W[y.cw[0] + i];
        }
    }
    fxt_sync_ws<WSAS>(); FXT_STAMP(1);

    WsCF feat = nullptr;            // input of the dense stack when it is not the one-hot
    if (n.kind == 0) {
        const int L1 = n.L1, K = n.K;
        WsF a1 = ws + w.a[0]; WsF a2 = ws + w.a[1]; WsF a3 = ws + w.a[2];
        // conv1 ('valid') on a one-hot input: a sum of K kernel rows
        if constexpr (MODE >= 2) {
            if (conv1_staged) {
                FXT_FOR(i, R * L1 * F, wg) {
                    const int o = i % F, rt = i / F, t = rt % L1, r = rt / L1;
                    float s = wbuf[K * A * F + o];
                    for (int jj = 0; jj < K; ++jj) s += wbuf[(jj * A + codes[r * L + t + jj]) * F + o];
                    a1[fxt_xi<LAY>(rt, o, ldF)] = s > 0.f ? s : 0.f;
                }
            }
        }
        if (MODE < 2 || !conv1_staged)
        FXT_FOR(i, R * L1 * F, wg) {
            const int o = i % F, rt = i / F, t = rt % L1, r = rt / L1;
            float s = W[y.cb[0] + o];
            for (int jj = 0; jj < K; ++jj) s += W[y.cw[0] + (jj * A + codes[r * L + t + jj]) * ldw + o];
            a1[fxt_xi<LAY>(rt, o, ldF)] = s > 0.f ? s : 0.f;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(2);
        {   // conv2 ('same', K taps)
            WCF b = W + y.cb[1];
            struct Put { WsF y; WCF b; int ld; FXT_HD void put(int m, int nn, float v) const { v += b[nn]; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            struct Put3 { WsF y; WCF b; int ld; FXT_HD float pre(int nn) const { return b[nn]; }
                          FXT_HD void put(int m, int nn, float v, float bias) const { v += bias; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            if constexpr (MODE == 3)
                fxt_conv32_staged<WSAS, WAS, 0>(wg, R * L1, K, FxtConvAZ<WsCF, WsCF4>{a1, L1, (K - 1) / 2, w.zero - w.a[0], dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, Put3{a2, b, ldF}, W + y.cw[1], wbuf, G3, tap, true, false, W + y.cw[2], n.K3);
            else if constexpr (MODE == 2)
                fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K, F, ConvA{a1, L1, ldF, (K - 1) / 2, dL1}, FxtConvW<WsCF>{wbuf, F, fxt_ld_w(F)}, Put{a2, b, ldF},
                                      W + y.cw[1], wbuf, stage_taps, F, F, fxt_ld_w(F));
            else
            fxt_gemm(wg, R * L1, F, K, F, ConvA{a1, L1, ldF, (K - 1) / 2, dL1}, FxtConvW<WCF>{W + y.cw[1], F, ldw}, Put{a2, b, ldF}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(3);
        {   // conv3 ('same', A - 1 taps)
            WCF b = W + y.cb[2];
            struct Put { WsF y; WCF b; int ld; FXT_HD void put(int m, int nn, float v) const { v += b[nn]; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            struct Put3 { WsF y; WCF b; int ld; FXT_HD float pre(int nn) const { return b[nn]; }
                          FXT_HD void put(int m, int nn, float v, float bias) const { v += bias; y[fxt_xi<LAY>(m, nn, ld)] = v > 0.f ? v : 0.f; } };
            if constexpr (MODE == 3)
                fxt_conv32_staged<WSAS, WAS, 0>(wg, R * L1, n.K3, FxtConvAZ<WsCF, WsCF4>{a2, L1, (n.K3 - 1) / 2, w.zero - w.a[1], dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, Put3{a3, b, ldF}, W + y.cw[2], wbuf, G3, tap, true, false, (WCF) nullptr, 0,
                                                [&](int k) {           // profiling aid (train_trace): group 0 of conv3's forward -- every wave's clock at the end of its MFMAs (44 + wave); wave 0's at the group's start (60) and at the next group's (61)
#if FXT_DEVICE
                                                    if (j.dbg && slice == 0 && (wg.tid & 63) == 0) {
                                                        if (k == 1) j.dbg[44 + (wg.tid >> 6)] = wall_clock64();
                                                        if (k == 0 && wg.tid == 0) j.dbg[60] = wall_clock64();
                                                        if (k == 2 && wg.tid == 0) j.dbg[61] = wall_clock64();
                                                    }
#endif
                                                    (void)k; });
            else if constexpr (MODE == 2)
                fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, n.K3, F, ConvA{a2, L1, ldF, (n.K3 - 1) / 2, dL1}, FxtConvW<WsCF>{wbuf, F, fxt_ld_w(F)}, Put{a3, b, ldF},
                                      W + y.cw[2], wbuf, stage_taps, F, F, fxt_ld_w(F));
            else
            fxt_gemm(wg, R * L1, F, n.K3, F, ConvA{a2, L1, ldF, (n.K3 - 1) / 2, dL1}, FxtConvW<WCF>{W + y.cw[2], F, ldw}, Put{a3, b, ldF}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(4);
        WsF g = ws + w.g; WsF cnt = ws + w.cnt;
        bool pooled = false;
        if constexpr (MODE != 0) {
            // The long sequences these modes serve run ONE row per slice: a thread per (row, channel) leaves 32 of 1024 threads with two
            // dependent walks over 233 positions (~20 us of a step).  Here every (row, channel) is shared by FXT_POOL_PARTS threads,
            // position t to thread t mod PARTS; partial maxima and tie counts meet in the (still unused) gradient array dzB.  A maximum
            // and an integer count do not depend on the order they are taken in; the one order-dependent case of the walk below -- a NaN
            // in position 0 stays, a NaN elsewhere is skipped -- is kept: the same bits.
            constexpr int PARTS = 32;
            if (L1 >= 2 * PARTS) {
                pooled = true;
                WsF pmax = ws + w.dzB; WsF pcnt = pmax + R * F * PARTS;        // (2 R F PARTS <= R L1 F floats)
                FXT_FOR(i, R * F * PARTS, wg) {
                    const int part = i % PARTS, rf = i / PARTS, r = rf / F, f = rf - r * F;
                    float mx = -INFINITY;
                    for (int t = part; t < L1; t += PARTS) { const float v = a3[fxt_xi<LAY>(r * L1 + t, f, ldF)]; mx = v > mx ? v : mx; }
                    pmax[i] = mx;
                }
                fxt_sync_ws<WSAS>();
                FXT_FOR(i, R * F * PARTS, wg) {
                    const int part = i % PARTS, rf = i / PARTS, r = rf / F, f = rf - r * F;
                    float mx = pmax[rf * PARTS];
                    for (int q = 1; q < PARTS; ++q) { const float v = pmax[rf * PARTS + q]; mx = v > mx ? v : mx; }
                    const float first = a3[fxt_xi<LAY>(r * L1, f, ldF)];
                    if (first != first) mx = first;
                    int c = 0;
                    for (int t = part; t < L1; t += PARTS) c += a3[fxt_xi<LAY>(r * L1 + t, f, ldF)] == mx;
                    pcnt[i] = (float)c;
                    if (part == 0) g[r * ldF + f] = mx;
                }
                fxt_sync_ws<WSAS>();
                FXT_FOR(i, R * F, wg) {
                    const int r = i / F, f = i - r * F;
                    float c = 0.f;
                    for (int q = 0; q < PARTS; ++q) c += pcnt[i * PARTS + q];      // (small integers: exact in any order)
                    cnt[r * ldF + f] = c;
                }
            }
        }
        if (!pooled)
        FXT_FOR(i, R * F, wg) {             // GlobalMaxPooling1D + the number of positions that attain the maximum
            const int r = i / F, f = i - r * F;
            float mx = a3[fxt_xi<LAY>(r * L1, f, ldF)];
            for (int t = 1; t < L1; ++t) { const float v = a3[fxt_xi<LAY>(r * L1 + t, f, ldF)]; mx = v > mx ? v : mx; }
            int c = 0;
            for (int t = 0; t < L1; ++t) c += a3[fxt_xi<LAY>(r * L1 + t, f, ldF)] == mx;
            g[r * ldF + f] = mx; cnt[r * ldF + f] = (float)c;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(5);
        feat = g;
    }

    // ---- dense stack, forward  (unrolled over the at most four layers: with a compile-time layer index the workspace
    // offsets are registers -- indexed at run time the little offset table lived in scratch memory, a global round trip
    // per access in the middle of the step)
#if FXT_DEVICE
#pragma unroll
#endif
    for (int li = 0; li < FXT_MAX_LAYERS; ++li) {
        if (li >= n.nl) break;
        const int Kd = n.dim[li], Nd = n.dim[li + 1];
        WCF Wl = W + y.w[li];
        WCF bl = W + y.b[li];
        WsF out = ws + w.act[li];
        const int ld_in = (li == 0 && !n.onehot_in) ? ldF : Kd;     // row stride of the layer's input
        const bool last = li == n.nl - 1;
        const bool drop = li == n.drop_layer;
        if (li == 0 && n.onehot_in) {
            FXT_FOR(i, R * Nd, wg) {        // one-hot input: sum of L rows
                const int r = i / Nd, o = i - r * Nd;
                float s = bl[o];
                for (int l = 0; l < L; ++l) s += Wl[(l * A + codes[r * L + l]) * Nd + o];
                out[i] = (last || s > 0.f) ? s : 0.f;
            }
        } else {
            WsCF in = li == 0 ? feat : ws + w.act[li - 1];
            struct Put {
                WsF y; WCF b; int Nd; bool last, drop; const FxtJob* j; int step, slot0; float ks;
                FXT_HD void put(int m, int nn, float v) const {
                    v += b[nn];
                    if (!last) v = v > 0.f ? v : 0.f;
                    if (drop) v = fxt_keep(*j, step, slot0 + m, nn) ? v * ks : 0.f;
                    y[m * Nd + nn] = v;
                }
            };
            fxt_gemm(wg, R, Nd, 1, Kd, FxtRowMajorA<WsCF>{in, ld_in}, FxtRowMajorB<WCF>{Wl, Nd}, Put{out, bl, Nd, last, drop, &j, step, slot0, keep_scale}, 0, split);
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(20 + li);
    }

    // ---- loss: d(mean over valid rows of (pred - y)^2) / d pred
    {
        // (selects, not w.act[n.nl - 1]: a run-time index would put the offset table into scratch memory)
        const int last_act = n.nl == 4 ? w.act[3] : (n.nl == 3 ? w.act[2] : (n.nl == 2 ? w.act[1] : w.act[0]));
        const int last_du = n.nl == 4 ? w.du[3] : (n.nl == 3 ? w.du[2] : (n.nl == 2 ? w.du[1] : w.du[0]));
        WsCF pred = ws + last_act;
        WsF du = ws + last_du;
        FXT_FOR(r, R, wg) {
            const float e = ws[w.yvalid + r] != 0.f ? pred[r] - ws[w.ylab + r] : 0.f;
            du[r] = 2.f * e / (float)nvalid;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(7);
        FXT_FOR(i, 1, wg) {                 // the slice's sum of squared errors (fixed order)
            float sse = 0.f;
            for (int r = 0; r < R; ++r) { const float e = du[r] * (float)nvalid * 0.5f; sse += e * e; }
#if FXT_DEVICE
            if (j.agent_io) fxt_store_agent(&part[n.P], sse); else
#endif
            part[n.P] = sse;
        }
    }

    // ---- dense stack, backward
#if FXT_DEVICE
#pragma unroll
#endif
    for (int lq = 0; lq < FXT_MAX_LAYERS; ++lq) {
        const int li = FXT_MAX_LAYERS - 1 - lq;
        if (li >= n.nl) continue;
        const int Kd = n.dim[li], Nd = n.dim[li + 1];
        WCF Wl = W + y.w[li];
        WsCF du = ws + w.du[li];
        const int ld_in = (li == 0 && !n.onehot_in) ? ldF : Kd;     // row stride of the layer's input
        struct PutW {
            float* gw; float* gb; int Kd, Nd; bool agent;
            FXT_HD void put(int m, int nn, float v) const {
                float* p = m < Kd ? gw + m * Nd + nn : gb + nn;
#if FXT_DEVICE
                if (agent) { fxt_store_agent(p, v); return; }
#endif
                *p = v;
            }
        };
        const PutW putw{part + n.off_w[li], part + n.off_b[li], Kd, Nd, j.agent_io != 0};
        if (li == 0 && n.onehot_in) {
            // (one thread per output element -- 8 FMAs each, no per-tile bookkeeping -- was measured SLOWER than the 49
            // two-k-step MFMA tiles here: 7.4 vs 6.1 us for the 100 x 100 layer, profiles/r3_train_trace.log)
            fxt_gemm(wg, Kd + 1, Nd, 1, R, FxtOneHotWGradA<WsCI>{codes, L, A, Kd, 0, dA}, FxtRowMajorB<WsCF>{du, Nd}, putw);
        } else {
            WsCF in = li == 0 ? feat : ws + w.act[li - 1];
            // gradient w.r.t. the layer's input FIRST (few tiles, Nd k-steps each); through the previous layer's ReLU (and
            // Dropout: a dropped unit's stored output is 0, a kept one carries the 1 / (1 - rate) scale) ...
            if (li > 0) {
                const bool dropped = (li - 1) == n.drop_layer;
                struct PutX { WsF d; WsCF y; int Kd; float ks; FXT_HD void put(int m, int nn, float v) const { d[m * Kd + nn] = y[m * Kd + nn] > 0.f ? v * ks : 0.f; } };
                fxt_gemm(wg, R, Kd, 1, Nd, FxtRowMajorA<WsCF>{du, Nd}, FxtTransB<WCF>{Wl, Nd}, PutX{ws + w.du[li - 1], in, Kd, dropped ? keep_scale : 1.f}, 0, split);
            } else {
                struct PutG { WsF d; int ld; FXT_HD void put(int m, int nn, float v) const { d[m * ld + nn] = v; } };
                fxt_gemm(wg, R, Kd, 1, Nd, FxtRowMajorA<WsCF>{du, Nd}, FxtTransB<WCF>{Wl, Nd}, PutG{ws + w.dg, ld_in}, 0, split);
            }
            // ... then the weight gradient (many tiles of R / 4 k-steps), dealt on from the wave behind the last long tile
            fxt_gemm(wg, Kd + 1, Nd, 1, R, FxtDenseWGradA<WsCF>{in, Kd, ld_in}, FxtRowMajorB<WsCF>{du, Nd}, putw, fxt_jobs(R, Kd, 1, Nd, nwv, can_split));
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(30 + li);
    }

    if (n.kind == 0) {
        const int L1 = n.L1, K = n.K, K3 = n.K3;
        WsCF a1 = ws + w.a[0]; WsCF a2 = ws + w.a[1]; WsCF a3 = ws + w.a[2];
        WsCF g = ws + w.g; WsCF cnt = ws + w.cnt; WsCF dg = ws + w.dg;
        WsF dzA = ws + w.dzA; WsF dzB = ws + w.dzB;
        if constexpr (MODE == 3) tap.template fetch<1>(wg, W + y.cw[2], 0, G3 < K3 ? G3 : K3);     // conv3's first group for the input gradient, behind this phase
        FXT_FOR(i, R * L1 * F, wg) {        // max-pool backward (ties share evenly) through conv3's ReLU
            const int f = i % F, rt = i / F, r = rt / L1;
            const float v = a3[fxt_xi<LAY>(rt, f, ldF)];
            dzA[fxt_xi<LAY>(rt, f, ldF)] = (v > 0.f && v == g[r * ldF + f]) ? dg[r * ldF + f] / cnt[r * ldF + f] : 0.f;
        }
        fxt_sync_ws<WSAS>(); FXT_STAMP(9);
        struct PutW {
            float* gw; float* gb; int rows, F; bool agent;
            FXT_HD void put(int m, int nn, float v) const {
                float* p = m < rows ? gw + m * F + nn : gb + nn;
#if FXT_DEVICE
                if (agent) { fxt_store_agent(p, v); return; }
#endif
                *p = v;
            }
        };
        const bool ag = j.agent_io != 0;
        struct PutX { WsF d; WsCF y; int ld; FXT_HD void put(int m, int nn, float v) const { d[fxt_xi<LAY>(m, nn, ld)] = y[fxt_xi<LAY>(m, nn, ld)] > 0.f ? v : 0.f; } };
        struct PutX3 { WsF d; WsCF y; int ld; FXT_HD int pre(int) const { return 0; }
                       FXT_HD void put(int m, int nn, float v, int) const { d[fxt_xi<LAY>(m, nn, ld)] = y[fxt_xi<LAY>(m, nn, ld)] > 0.f ? v : 0.f; } };
        // conv3: input gradient (few tiles, K3 x F / 4 k-steps) first, the weight gradient dealt on behind it
        if constexpr (MODE == 3)
            fxt_conv32_staged<WSAS, WAS, 1>(wg, R * L1, K3, FxtConvGradAZ<WsCF, WsCF4>{dzA, L1, (K3 - 1) / 2, w.zero - w.dzA, dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, PutX3{dzB, a2, ldF}, W + y.cw[2], wbuf, G3, tap, true, false,
                                            W + y.cw[1], K);          // (leaves conv2's first group in `tap`: fetched BEFORE this phase's partial stores)
        else if constexpr (MODE == 2)
            fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K3, F, ConvGradA{dzA, L1, ldF, (K3 - 1) / 2, dL1}, FxtConvGradW<WsCF>{wbuf, F, fxt_ld_w(F)}, PutX{dzB, a2, ldF},
                                  W + y.cw[2], wbuf, stage_taps, F, F, fxt_ld_w(F));
        else
        fxt_gemm(wg, R * L1, F, K3, F, ConvGradA{dzA, L1, ldF, (K3 - 1) / 2, dL1}, FxtConvGradW<WCF>{W + y.cw[2], F, ldw}, PutX{dzB, a2, ldF}, 0, split);
        if constexpr (MODE == 3)
        {   FXT_STAMP(40);
            // the hook: conv2's group goes into the staging buffer (every wave is through with conv3's last group behind the barrier)
            // before this phase's 78 KiB of partial stores are issued; conv2's input gradient then starts without touching global memory
            auto commit = [&]() { fxt_sync_ws<WSAS>(); tap.template store<WSAS>(wg, wbuf); FXT_STAMP(41); };
            fxt_conv32_wgrad<D::fixed ? D::A - 1 : 0>(wg, R, L1, K3, (K3 - 1) / 2, a2, (WsCF)dzA, (WsCF)(ws + w.zero), PutW{part + n.off_cw[2], part + n.off_cb[2], K3 * F, F, ag}, commit);
        }
        else
        fxt_gemm(wg, K3 * F + 1, F, R, L1, ConvWGradA{a2, L1, F, ldF, (K3 - 1) / 2, K3 * F, dF}, PosMajorB{dzA, L1, ldF}, PutW{part + n.off_cw[2], part + n.off_cb[2], K3 * F, F, ag}, fxt_jobs(R * L1, F, K3, F, nwv, can_split));
        fxt_sync_ws<WSAS>(); FXT_STAMP(10);
        // conv2
        if constexpr (MODE == 3)
            fxt_conv32_staged<WSAS, WAS, 1>(wg, R * L1, K, FxtConvGradAZ<WsCF, WsCF4>{dzB, L1, (K - 1) / 2, w.zero - w.dzB, dL1}, FxtConvW4<WsCF, WsCF4>{wbuf}, PutX3{dzA, a1, ldF}, W + y.cw[1], wbuf, G3, tap, false, true);
        else if constexpr (MODE == 2)
            fxt_gemm_staged<WSAS, WAS>(wg, R * L1, F, K, F, ConvGradA{dzB, L1, ldF, (K - 1) / 2, dL1}, FxtConvGradW<WsCF>{wbuf, F, fxt_ld_w(F)}, PutX{dzA, a1, ldF},
                                  W + y.cw[1], wbuf, stage_taps, F, F, fxt_ld_w(F));
        else
        fxt_gemm(wg, R * L1, F, K, F, ConvGradA{dzB, L1, ldF, (K - 1) / 2, dL1}, FxtConvGradW<WCF>{W + y.cw[1], F, ldw}, PutX{dzA, a1, ldF}, 0, split);
        if constexpr (MODE == 3)
        {   FXT_STAMP(42);
            fxt_conv32_wgrad<D::fixed ? D::K : 0>(wg, R, L1, K, (K - 1) / 2, a1, (WsCF)dzB, (WsCF)(ws + w.zero), PutW{part + n.off_cw[1], part + n.off_cb[1], K * F, F, ag});
            FXT_STAMP(43);
        }
        else
        fxt_gemm(wg, K * F + 1, F, R, L1, ConvWGradA{a1, L1, F, ldF, (K - 1) / 2, K * F, dF}, PosMajorB{dzB, L1, ldF}, PutW{part + n.off_cw[1], part + n.off_cb[1], K * F, F, ag}, fxt_jobs(R * L1, F, K, F, nwv, can_split));
        fxt_sync_ws<WSAS>(); FXT_STAMP(11);
        // conv1 (one-hot input, 'valid')
        if constexpr (MODE == 3)
            fxt_gemm(wg, K * A + 1, F, R, L1, FxtOneHotWGradA<WsCI>{codes, L, A, K * A, 1, dA}, FxtPosMajorBL<WsCF, 2>{dzA, L1, ldF}, PutW{part + n.off_cw[0], part + n.off_cb[0], K * A, F, ag});
        else
        fxt_gemm(wg, K * A + 1, F, R, L1, FxtOneHotWGradA<WsCI>{codes, L, A, K * A, 1, dA}, PosMajorB{dzA, L1, ldF}, PutW{part + n.off_cw[0], part + n.off_cb[0], K * A, F, ag});
    }
    FXT_STAMP(63);
}

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
