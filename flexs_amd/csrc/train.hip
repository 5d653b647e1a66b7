// fx_train_fit: `KerasModel.train` (flexs/baselines/models/keras_model.py:49-67 -> model.fit(batch_size, epochs)) for
// one or several ensemble members at once, hand-written for gfx950 (train_core.h).
//
// One mini-batch step = TWO launches whatever the number of members:
//   k_train_fb    grid (slices, members): forward + backward of R mini-batch rows, per-slice gradient partials;
//   k_train_adam  grid (parameter blocks, members): fixed-order sum of the partials + Keras-Adam update.
// A fit of `epochs` x ceil(n / batch) steps enqueues them back to back on the engine's stream from C -- no Python, no
// framework dispatch between steps -- with everything the steps need resident on the device: the data set as bytes,
// the labels, every step's mini-batch composition (the host draws the shuffles, as Keras does), per-step learning
// rates.  Weights and optimiser moments go in and come out as plain host arrays in Keras get_weights() order, so
// the caller (flexs_amd/training.py) keeps them on the `Architecture` between explorer rounds
// (flexs/explorer.py:157-160) exactly as before.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "fx_common.h"
#include "mfma_common.h"
#include "train_core.h"

namespace {

constexpr int FB_MAX_THREADS = 1024;
constexpr size_t FB_LDS_BUDGET = 150 * 1024;

__global__ void __launch_bounds__(FB_MAX_THREADS) k_train_fb(const FxtJob* __restrict__ jobs, int step, const uint8_t* __restrict__ ascii,
                                                         const uint8_t* __restrict__ lut, const float* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) float fxt_smem[];
    const FxtJob& j = jobs[blockIdx.y];
    if (step >= j.total_steps || (int)blockIdx.x >= j.S) return;
    typedef FxtMem<3>::F lds_f;             // (plain pointers in the host pass of this file, address-space-qualified on the device)
    typedef FxtMem<3>::CF lds_cf;
    typedef FxtMem<1>::F glb_f;
    typedef FxtMem<1>::CF glb_cf;
    const FxtWg wg{(int)threadIdx.x, (int)blockDim.x};
    const int slice = (int)blockIdx.x;
    FXT_STAMP(62);                          // kernel entry (before the weights are staged)
    if (j.w_in_lds) {
        // the member's whole parameter vector next to the slice's workspace: every operand of every layer then comes
        // from LDS through ds_read (see train_core.h "address spaces")
        float* wl = fxt_smem + j.ws_slice;
        typedef float v4f __attribute__((ext_vector_type(4)));
        // (the image has padded conv-kernel rows -- train_core.h "Row strides": a 16-byte piece never straddles a row)
        const FxtLay lay = fxt_lay(j.net, true);
        const int n4 = j.net.P >> 2;
        const v4f* src = reinterpret_cast<const v4f*>(j.w);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 12 * blockDim.x) {      // 12 x 16 bytes in flight per thread: ~2 round trips for 100 KiB
            v4f v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int i = i0 + k * (int)blockDim.x; v[k] = src[i < n4 ? i : n4 - 1]; }   // (clamped: the loads stay unconditional, in registers)
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int i = i0 + k * (int)blockDim.x; if (i < n4) *reinterpret_cast<v4f*>(wl + fxt_image_off(j.net, lay, 4 * i)) = v[k]; }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < j.net.P; i += blockDim.x) wl[fxt_image_off(j.net, lay, i)] = j.w[i];
        // (published by the first fxt_sync of the step)
        const lds_f sp33 = j.split_off ? (lds_f)(fxt_smem + j.split_off) : (lds_f) nullptr;     // split-K scratch behind the weights
        // canonical shapes: the same source with the dimensions as compile-time constants (train_core.h FxtDims)
        switch (j.canon) {
            case 1: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
            case 2: fxt_forward_backward<3, 3, FxtDims<1, 4, 0, 100, 0, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
            case 3: fxt_forward_backward<3, 3, FxtDims<2, 20, 0, 100, 0, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
            case 4: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;    // TF-binding
            case 5: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8, 14>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;   // RNA L = 14
            case 6: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 4, 14>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;   // RNA L = 14, four rows per slice (the weights fit beside them)
            case 7: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 0>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;       // the 4-letter CNN at any length and rows per slice
            default: fxt_forward_backward<3, 3>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33);
        }
    } else if (j.ws_in_lds) {
        fxt_forward_backward<3, 1>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (glb_cf)j.w, j.split_off ? (lds_f)(fxt_smem + j.split_off) : (lds_f) nullptr);
    } else {
        fxt_forward_backward<1, 1>(j, wg, step, slice, ascii, lut, labels, (glb_f)(j.ws + (long long)slice * j.ws_slice), (glb_cf)j.w);
    }
}

// The same step for fits whose members ALL store their position-major arrays with rotated rows (train_core.h "Rotated rows",
// FxtJob::canon = -1: workspace in LDS, weights from L2).  A kernel of its own, sharing no instantiation with k_train_fb, so that
// the kernel every other fit runs keeps its code instruction for instruction.
__global__ void __launch_bounds__(FB_MAX_THREADS) k_train_fb_swz(const FxtJob* __restrict__ jobs, int step, const uint8_t* __restrict__ ascii,
                                                             const uint8_t* __restrict__ lut, const float* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) float fxt_smem[];
    const FxtJob& j = jobs[blockIdx.y];
    if (step >= j.total_steps || (int)blockIdx.x >= j.S) return;
    const FxtWg wg{(int)threadIdx.x, (int)blockDim.x};
    if (j.canon == -2)       // + the gradient array over the last conv output, conv kernels staged through LDS (j.split_off taps at a time)
        fxt_forward_backward<3, 1, FxtDimsAny, 2>(j, wg, step, (int)blockIdx.x, ascii, lut, labels, (FxtMem<3>::F)fxt_smem, (FxtMem<1>::CF)j.w, (FxtMem<3>::F) nullptr);
    else
        fxt_forward_backward<3, 1, FxtDimsAny, 1>(j, wg, step, (int)blockIdx.x, ascii, lut, labels, (FxtMem<3>::F)fxt_smem, (FxtMem<1>::CF)j.w, (FxtMem<3>::F) nullptr);
}

// Round 5: the F = 32 protein CNNs (train_core.h "MODE 3": paired tiles over conflict-free rotated kernel rows, register-prefetched
// staging, sliding-window weight gradient).  FxtJob::canon = -3; again a kernel of its own (its register budget is its own).
__global__ void __launch_bounds__(FB_MAX_THREADS) k_train_fb_c32(const FxtJob* __restrict__ jobs, int step, const uint8_t* __restrict__ ascii,
                                                             const uint8_t* __restrict__ lut, const float* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) float fxt_smem[];
    const FxtJob& j = jobs[blockIdx.y];
    if (step >= j.total_steps || (int)blockIdx.x >= j.S) return;
    const FxtWg wg{(int)threadIdx.x, (int)blockDim.x};
    fxt_forward_backward<3, 1, FxtDimsAny, 3>(j, wg, step, (int)blockIdx.x, ascii, lut, labels, (FxtMem<3>::F)fxt_smem, (FxtMem<1>::CF)j.w, (FxtMem<3>::F) nullptr);
}
// ... and the protein surrogate of the BASELINE configs -- CNN(32, 100, kernel 5) on 20 letters, one row per slice -- with its dimensions
// as constants (train_core.h FxtDims: same bits; 1/3 of the code -- the shape-agnostic body is 90 KiB, streamed through the 64 KiB
// instruction cache by every workgroup: 643 -> 483 us per launch at L = 237).  FxtJob::canon = -4, when ALL members have this shape.
__global__ void __launch_bounds__(FB_MAX_THREADS) k_train_fb_c32p(const FxtJob* __restrict__ jobs, int step, const uint8_t* __restrict__ ascii,
                                                              const uint8_t* __restrict__ lut, const float* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) float fxt_smem[];
    const FxtJob& j = jobs[blockIdx.y];
    if (step >= j.total_steps || (int)blockIdx.x >= j.S) return;
    const FxtWg wg{(int)threadIdx.x, (int)blockDim.x};
    if (j.R == 1)
        fxt_forward_backward<3, 1, FxtDims<0, 20, 32, 100, 5, 1>, 3>(j, wg, step, (int)blockIdx.x, ascii, lut, labels, (FxtMem<3>::F)fxt_smem, (FxtMem<1>::CF)j.w, (FxtMem<3>::F) nullptr);
    else     // short protein sequences (fewer than 36 residues: several rows per slice) -- the same constants, the rows a run-time value
        fxt_forward_backward<3, 1, FxtDims<0, 20, 32, 100, 5, 0>, 3>(j, wg, step, (int)blockIdx.x, ascii, lut, labels, (FxtMem<3>::F)fxt_smem, (FxtMem<1>::CF)j.w, (FxtMem<3>::F) nullptr);
}

__global__ void __launch_bounds__(256) k_train_adam(const FxtJob* __restrict__ jobs, int step) {
    const FxtJob& j = jobs[blockIdx.y];
    if (step >= j.total_steps) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (j.pstride && (j.pstride & 3) == 0) {                // aligned rows (fx_train_fit's arena): four parameters per thread, 16-byte loads
        const int i4 = 4 * i;
        if (i4 + 3 < j.net.P) fxt_adam4(j, step, i4);
        else for (int k = i4; k < j.net.P && k < i4 + 4; ++k) fxt_adam(j, step, k);
    } else {
        // unaligned rows (pstride 0 = P + 1 is legal in FxtJob; fx_train_fit never builds one): one parameter at a time, grid-stride --
        // the grid is sized for four per thread
        for (int k = i; k < j.net.P; k += (int)(gridDim.x * blockDim.x)) fxt_adam(j, step, k);
    }
    if (blockIdx.x == 0 && j.step_loss) {
        // the step's loss = the slices' squared-error sums added in slice order.  One thread walking S dependent-looking global
        // loads was the kernel's critical path (256 slices at one row per slice: ~75 us whatever the rest of the grid did, round 5):
        // the workgroup fetches them side by side into LDS, one thread adds them in the same order
        __shared__ float sse_s[512];
        const int sidx = step % j.steps_per_epoch;
        const int nvalid = (j.n - sidx * j.batch) < j.batch ? (j.n - sidx * j.batch) : j.batch;
        const long long ps = fxt_pstride(j);
        float sse = 0.f;
        for (int c0 = 0; c0 < j.S; c0 += 512) {
            const int cn = j.S - c0 < 512 ? j.S - c0 : 512;
            __syncthreads();
            for (int k = threadIdx.x; k < cn; k += blockDim.x) sse_s[k] = j.partial[(long long)(c0 + k) * ps + j.net.P];
            __syncthreads();
            if (threadIdx.x == 0) for (int k = 0; k < cn; ++k) sse += sse_s[k];
        }
        if (threadIdx.x == 0) j.step_loss[step] = sse / (float)nvalid;
    }
}

// ---- the whole fit as ONE launch (round 4) ----------------------------------------------------------------------------------
// k_train_fb + k_train_adam per mini-batch step are two dependent launches -- 160 for a fit of 1000 sequences -- and every
// forward+backward workgroup re-stages its member's weights from L2 at its start.  Here the (slices x members) workgroups stay
// for the whole fit: per step a workgroup runs its slice's forward + backward (the same fxt_forward_backward: same bits), the S
// workgroups of a MEMBER meet at a barrier in device memory (members never wait for each other), each then sums the partials
// of ITS share of the parameters in slice order and applies Keras' Adam to them (the same fxt_adam: same bits), a second
// barrier, and the new weights are read back into LDS.  Cross-workgroup visibility: the partials / weights are plain stores,
// published by an agent-scope release (L2 write-back) before the arrival and picked up behind an agent-scope acquire after
// the barrier -- twice per step, not per work unit.  Needs all workgroups co-resident (checked by the host through the
// occupancy API; the resident scoring generation is told to leave first); a barrier that is not passed within ~2 s raises
// the abort word and everybody leaves (the host reports FX_ESTATE) instead of hanging the device.
struct FxtBar { unsigned count[64]; unsigned abort; };

__device__ __forceinline__ bool fxt_member_barrier(FxtBar* bar, int m, unsigned target, int tid) {
    __shared__ int s_abort;
    fx_wait_vm(0);                                         // this wave's write-through stores (partials / weights) have been performed
    __syncthreads();
    if (tid == 0) {
        // no fence: what the workgroups exchange is written through and read past the non-coherent cache levels (agent_io)
        __hip_atomic_fetch_add(&bar->count[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        int ab = 0;
        while (__hip_atomic_load(&bar->count[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(&bar->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
            if (wall_clock64() - t0 > 200000000ull) {      // 2 s at 100 MHz: somebody never arrived (not co-resident?)
                __hip_atomic_store(&bar->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ab = 1;
                break;
            }
        }
        s_abort = ab;
    }
    __syncthreads();
    return s_abort == 0;
}

__global__ void __launch_bounds__(FB_MAX_THREADS) k_train_fit(const FxtJob* __restrict__ jobs, const uint8_t* __restrict__ ascii,
                                                          const uint8_t* __restrict__ lut, const float* __restrict__ labels, FxtBar* bar) {
    extern __shared__ __attribute__((aligned(16))) float fxt_smem[];
    const FxtJob& j = jobs[blockIdx.y];
    if ((int)blockIdx.x >= j.S) return;                    // (a member with fewer slices than the grid is wide: not part of its barrier)
    typedef FxtMem<3>::F lds_f;
    typedef FxtMem<3>::CF lds_cf;
    typedef FxtMem<1>::F glb_f;
    typedef FxtMem<1>::CF glb_cf;
    const FxtWg wg{(int)threadIdx.x, (int)blockDim.x};
    const int slice = (int)blockIdx.x, m = (int)blockIdx.y;
    const int P = j.net.P;
    const int p_lo = (int)((long long)P * slice / j.S), p_hi = (int)((long long)P * (slice + 1) / j.S);
    float* wl = fxt_smem + j.ws_slice;
    auto stage_weights = [&]() {
        // the member's weights, as the workgroups that own their shares wrote them through: read past the non-coherent levels,
        // eight 16-byte loads in flight per thread
        const int n4 = P >> 2;
        const f4* src = reinterpret_cast<const f4*>(j.w);
        const FxtLay lay = fxt_lay(j.net, true);           // (padded conv-kernel rows: train_core.h "Row strides")
        const int bd = (int)blockDim.x;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * bd) {
            f4 v[8];
            auto at = [&](int k) { const int i = i0 + k * bd; return src + (i < n4 ? i : n4 - 1); };
            fx_load16x8_agent(at(0), at(1), at(2), at(3), at(4), at(5), at(6), at(7), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + k * bd; if (i < n4) *reinterpret_cast<f4*>(wl + fxt_image_off(j.net, lay, 4 * i)) = v[k]; }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < P; i += blockDim.x) wl[fxt_image_off(j.net, lay, i)] = __hip_atomic_load(&j.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned phase = 0;
    for (int step = 0; step < j.total_steps; ++step) {
        if (j.w_in_lds) {
            stage_weights();                               // (published by the first fxt_sync of the step)
            const lds_f sp33 = j.split_off ? (lds_f)(fxt_smem + j.split_off) : (lds_f) nullptr;
            switch (j.canon) {                             // (the instantiations of k_train_fb: same bits)
                case 1: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 2: fxt_forward_backward<3, 3, FxtDims<1, 4, 0, 100, 0, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 3: fxt_forward_backward<3, 3, FxtDims<2, 20, 0, 100, 0, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 4: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8, 8>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 5: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 8, 14>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 6: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 4, 14>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                case 7: fxt_forward_backward<3, 3, FxtDims<0, 4, 32, 100, 5, 0>>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33); break;
                default: fxt_forward_backward<3, 3>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (lds_cf)wl, sp33);
            }
        } else if (j.ws_in_lds) {
            fxt_forward_backward<3, 1>(j, wg, step, slice, ascii, lut, labels, (lds_f)fxt_smem, (glb_cf)j.w, j.split_off ? (lds_f)(fxt_smem + j.split_off) : (lds_f) nullptr);
        } else {
            fxt_forward_backward<1, 1>(j, wg, step, slice, ascii, lut, labels, (glb_f)(j.ws + (long long)slice * j.ws_slice), (glb_cf)j.w);
        }
        if (!fxt_member_barrier(bar, m, (unsigned)j.S * ++phase, (int)threadIdx.x)) return;
        for (int i = p_lo + (int)threadIdx.x; i < p_hi; i += (int)blockDim.x) fxt_adam(j, step, i);
        if (slice == 0 && threadIdx.x == 0 && j.step_loss) fxt_step_loss(j, step);
        if (!fxt_member_barrier(bar, m, (unsigned)j.S * ++phase, (int)threadIdx.x)) return;
    }
}

// Rows per slice.  More slices = more workgroups (the machine has 256 CUs and a step of one small network is a few
// hundred thousand MACs), fewer rows per slice = emptier 16-row MFMA tiles in the dense layers and more partials to
// sum.  The choice depends on the member's OWN shape and batch size only -- never on how many members train in the
// same call -- so a member's gradient sums are cut the same way whether it trains alone, next to its ensemble, or on
// another rank (member-sharded training): the fit is bit-reproducible across those.
int rows_per_slice(const FxtNet& n, int batch, int num_cus, int forced) {
    if (forced > 0) return forced > 64 ? 64 : forced;
    int R = 16;
    while (R > 8 && (batch + R - 1) / R < num_cus / 8) R >>= 1;
    if (n.kind == 0 && n.L1 >= 32)          // long sequences: a row alone fills M tiles of the conv GEMMs
        while (R > 1 && (batch + R - 1) / R < num_cus) R >>= 1;
    return R;
}

struct Arena {                 // bump allocator over one device buffer (sizes first, then pointers)
    char* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += sizeof(T) * count;
        return p;
    }
};

}  // namespace

int fx_train_fit(fx_engine* e, fx_fit_job* jobs, int M, const uint8_t* ascii, int64_t n, int L, const uint8_t lut[256],
                 const float* labels) {
    if (!e) return FX_EINVAL;
    if (!jobs || M < 1 || M > 64 || !lut || n < 0 || L < 1) return fx_fail(e, FX_EINVAL, "fx_train_fit: bad arguments");
    const auto t_entry = std::chrono::steady_clock::now();
    auto since = [&]() { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(); };
    if (n == 0) return FX_OK;
    if (!ascii || !labels) return fx_fail(e, FX_EINVAL, "fx_train_fit: null data");
    if (n > (int64_t)1 << 30) return fx_fail(e, FX_EINVAL, "fx_train_fit: data set too large");
    fx_server_stop(e);                                     // the step kernels want every CU
    fx_lp_disarm(e);
    std::vector<FxtJob> hj((size_t)M);
    std::vector<std::vector<float>> lr((size_t)M);
    int max_steps = 0, max_S = 0, max_P = 0;
    int n_swz = 0, n_stage = 0, n_c32 = 0;                 // members eligible for rotated rows / staged conv kernels / the F = 32 form: used when ALL of the fit's members are
    std::vector<int> stage_taps((size_t)M, 0), c32_taps((size_t)M, 0);
    std::vector<FxtNet> c32_net((size_t)M);
    size_t lds_bytes = 0;
    for (int m = 0; m < M; ++m) {
        fx_fit_job& u = jobs[m];
        if (u.kind < FX_CNN || u.kind > FX_GE || u.L != L || u.A < 1 || u.A > 254 || u.H < 1 || u.batch < 1 || u.epochs < 0 ||
            !u.weights || !u.adam_m || !u.adam_v || !u.order || u.step < 0)
            return fx_fail(e, FX_EINVAL, "fx_train_fit: bad job");
        if (u.kind == FX_CNN && (u.F < 1 || u.K < 1 || u.A < 2 || L < u.K)) return fx_fail(e, FX_ESHAPE, "fx_train_fit: bad CNN shape");
        FxtJob& j = hj[(size_t)m];
        j = FxtJob{};
        j.net = fxt_net(u.kind, L, u.A, u.kind == FX_CNN ? u.F : 0, u.H, u.kind == FX_CNN ? u.K : 0);
        if ((int64_t)j.net.P != fx_num_params(FxShape{u.kind, L, u.A, j.net.F, u.H, j.net.K})) return fx_fail(e, FX_EINVAL, "fx_train_fit: parameter count mismatch");
        j.batch = u.batch;
        j.steps_per_epoch = (int)((n + u.batch - 1) / u.batch);
        j.total_steps = u.epochs * j.steps_per_epoch;
        j.n = (int)n;
        j.R = rows_per_slice(j.net, u.batch, e->num_cus, (int)e->train_rows);
        // A CNN whose weights would fit LDS beside a SMALLER slice runs fewer rows per slice (round 5): CNN(32, 100, 5) on 4 letters at
        // seq_len 14 -- the RNA landscapes -- needs 71 KiB of workspace at eight rows, 165 KiB with its 94 KiB weight image, and ran with
        // every B operand an L2 round trip (66 us per step; 36 KiB at four rows: image and workspace resident).  From the member's own
        // shape only, like R itself.
        if (j.net.kind == 0 && !e->train_rows && e->train_lds >= 2 && (j.net.F & 3) == 0) {
            const size_t img = (size_t)fxt_lay(j.net, true).total;
            if (img * 4 + 8192 <= FB_LDS_BUDGET)
                while (j.R > 2 && ((size_t)fxt_ws(j.net, j.R).total + img) * 4 > FB_LDS_BUDGET) j.R >>= 1;
        }
        j.S = (u.batch + j.R - 1) / j.R;
        j.seed = u.seed;
        j.pstride = (j.net.P + 1 + 31) & ~31;
        j.ws_slice = fxt_ws(j.net, j.R).total;
        if (j.net.kind == 0 && (size_t)j.ws_slice * 4 > FB_LDS_BUDGET) {     // padded rows just too large for LDS: unpadded in LDS beats padded in global memory
            FxtNet plain = j.net;
            plain.ldx = plain.F;
            const int total = fxt_ws(plain, j.R).total;
            if ((size_t)total * 4 <= FB_LDS_BUDGET) { j.net = plain; j.ws_slice = total; }
        }
        j.ws_in_lds = e->train_lds >= 1 && (size_t)j.ws_slice * 4 <= FB_LDS_BUDGET;
        const size_t w_image = (size_t)fxt_lay(j.net, true).total;     // the weights' LDS image (padded conv-kernel rows)
        j.w_in_lds = j.ws_in_lds && e->train_lds >= 2 && (j.net.F & 3) == 0 && ((size_t)j.ws_slice + w_image) * 4 <= FB_LDS_BUDGET;
        // split-K scratch (train_core.h fxt_gemm) behind the workspace (and the weights) when the LDS budget allows
        j.split_off = 0;
        if (j.ws_in_lds && e->train_split) {
            const size_t used = (((size_t)j.ws_slice + (j.w_in_lds ? w_image : 0)) + 3) & ~(size_t)3;
            if ((used + FXT_SPLIT_FLOATS) * 4 <= FB_LDS_BUDGET) j.split_off = (int)used;
        }
        if (j.ws_in_lds) lds_bytes = std::max(lds_bytes, ((size_t)j.ws_slice + (j.w_in_lds ? w_image : 0)) * 4);
        if (j.split_off) lds_bytes = std::max(lds_bytes, ((size_t)j.split_off + FXT_SPLIT_FLOATS) * 4);
        // canonical shapes get the instantiation with compile-time dimensions (workspace + weights in LDS, 8 rows per slice)
        j.canon = 0;
        if (e->train_canon && j.w_in_lds && j.R == 4 && u.kind == FX_CNN && u.A == 4 && u.F == 32 && u.H == 100 && u.K == 5 && L == 14 && j.net.ldx == fxt_ld_x(32)) j.canon = 6;
        if (e->train_canon && j.w_in_lds && j.R != 8 && j.canon == 0 && u.kind == FX_CNN && u.A == 4 && u.F == 32 && u.H == 100 && u.K == 5 && j.net.ldx == fxt_ld_x(32)) j.canon = 7;
        if (e->train_canon && j.w_in_lds && j.R == 8) {
            if (u.kind == FX_CNN && u.A == 4 && u.F == 32 && u.H == 100 && u.K == 5 && j.net.ldx == fxt_ld_x(32)) j.canon = L == 8 ? 4 : (L == 14 ? 5 : 1);
            if (u.kind == FX_MLP && u.A == 4 && u.H == 100) j.canon = 2;
            if (u.kind == FX_GE && u.A == 20 && u.H == 100) j.canon = 3;
        }
        // long protein CNNs (unpadded rows in LDS, weights from L2): rotated rows instead of 16-way conflicted ones (train_core.h)
        if (e->train_swizzle && j.net.kind == 0 && j.net.ldx == j.net.F && j.ws_in_lds && !j.w_in_lds && j.net.F >= 32 && (j.net.F & (j.net.F - 1)) == 0) {
            n_swz += 1;
            // train_swizzle = 2: how many taps of a conv kernel fit behind the four-array workspace (train_core.h MODE 2)
            const size_t ws2 = (size_t)fxt_ws(j.net, j.R, true).total, tap = (size_t)j.net.F * (size_t)fxt_ld_w(j.net.F);
            const int kmax = std::max(j.net.K, j.net.K3);
            int taps = ws2 * 4 < FB_LDS_BUDGET ? (int)std::min<size_t>((FB_LDS_BUDGET / 4 - ws2) / tap, (size_t)kmax) : 0;
            if (!fxt_staged_ok(j.R * j.net.L1, j.net.F, j.net.F, j.net.F, FB_MAX_THREADS / 64)) taps = 0;
            stage_taps[(size_t)m] = taps;
            if (e->train_swizzle >= 2 && taps >= 1) n_stage += 1;
        }
        // round 5, train_swizzle = 3: ANY CNN with 32 filters whose weights miss LDS (the 20-letter alphabets: 165 KiB) and whose
        // slice has at most 16 M tiles (R L1 <= 256) -- four rotated position-major arrays + at least one 4 KiB tap of staging in LDS;
        // reaches the sequences whose five-array layout misses the budget (L = 240 ... 260 at one row per slice) as well
        if (e->train_swizzle >= 3 && e->train_lds >= 1 && j.net.kind == 0 && j.net.F == 32 && !j.w_in_lds && j.net.K <= 4 * FXT_WG32_MAXT && j.net.K3 <= 4 * FXT_WG32_MAXT &&
            j.net.K3 >= 1 && fxt_conv32_ok(j.R * j.net.L1, j.net.F, FB_MAX_THREADS / 64)) {
            FxtNet rot = j.net;
            rot.ldx = rot.F;
            const size_t ws3 = (size_t)fxt_ws(rot, j.R, true).total;
            const int kmax = std::max(rot.K, rot.K3);
            const int taps = ws3 * 4 + 4096 <= FB_LDS_BUDGET ? (int)std::min<size_t>(std::min<size_t>((FB_LDS_BUDGET / 4 - ws3) / 1024, (size_t)kmax), (size_t)8) : 0;
            if (taps >= 1) { n_c32 += 1; c32_taps[(size_t)m] = taps; c32_net[(size_t)m] = rot; }
        }
        max_steps = std::max(max_steps, j.total_steps);
        max_S = std::max(max_S, j.S);
        max_P = std::max(max_P, j.net.P);
        lr[(size_t)m].resize((size_t)std::max(j.total_steps, 1));
        for (int s = 0; s < j.total_steps; ++s) {
            const double t = (double)(u.step + s + 1);           // keras Adam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)
            lr[(size_t)m][(size_t)s] = (float)(FXT_LR * std::sqrt(1.0 - std::pow(FXT_BETA_2, t)) / (1.0 - std::pow(FXT_BETA_1, t)));
        }
    }
    int threads = (int)e->train_threads;
    threads = threads >= 1024 ? 1024 : (threads >= 512 ? 512 : (threads >= 256 ? 256 : 1024));
    const bool c32 = n_c32 == M && threads == FB_MAX_THREADS;
    const bool any_swz = c32 || n_swz == M;
    const bool staged = !c32 && any_swz && n_stage == M && threads == FB_MAX_THREADS;     // (fxt_staged_ok was asked for 16 waves)
    bool c32p = c32 && e->train_canon != 0;                // every member the canonical protein shape?
    if (c32) {
        lds_bytes = 0;                                         // (the layouts sized above are not the ones these members run)
        for (int m = 0; m < M; ++m) c32p = c32p && c32_net[(size_t)m].A == 20 && c32_net[(size_t)m].H == 100 && c32_net[(size_t)m].K == 5;
        for (int m = 0; m < M; ++m) {
            FxtJob& j = hj[(size_t)m];
            j.net = c32_net[(size_t)m];
            j.canon = c32p ? -4 : -3;
            j.ws_in_lds = 1; j.w_in_lds = 0;
            j.split_off = c32_taps[(size_t)m];
            j.ws_slice = fxt_ws(j.net, j.R, true).total + c32_taps[(size_t)m] * 1024;
            lds_bytes = std::max(lds_bytes, (size_t)j.ws_slice * 4);
        }
    } else if (any_swz) {
        for (int m = 0; m < M; ++m) {
            FxtJob& j = hj[(size_t)m];
            j.canon = staged ? -2 : -1;
            j.split_off = staged ? stage_taps[(size_t)m] : 0;
            if (staged) {
                j.ws_slice = fxt_ws(j.net, j.R, true).total + stage_taps[(size_t)m] * j.net.F * fxt_ld_w(j.net.F);
                lds_bytes = std::max(lds_bytes, (size_t)j.ws_slice * 4);
            }
        }
    }
    for (int64_t i = 0; i < n * L; ++i)                          // alphabet.index raises ValueError (sequence_utils.py:46)
        if (lut[ascii[i]] == 0xFF) return fx_fail(e, FX_EBADCHAR, "character outside the alphabet in the training set");
    if (max_steps == 0) return FX_OK;
    FX_HIP(e, hipSetDevice(e->device));

    // ---- one device arena in three regions (round 4):
    //   A  per member: weights, Adam moments, per-step losses          -- uploaded AND downloaded
    //   B  job table, LUT, data, labels, per member: order, lr, masks   -- uploaded
    //   C  step barrier, per member: gradient partials, workspace       -- device only
    // A + B are filled in a pinned host image of the same layout and go up as ONE copy, A comes back as ONE copy: the 19 + 12
    // pageable hipMemcpyAsync calls of a three-member fit (each staged by the runtime, the downloads each a host wait) were
    // ~0.5 ms of a 4 ms Ensemble.train.
    size_t need = 0, bytes_a = 0, bytes_ab = 0;
    auto plan = [&](Arena& a) {
        for (int m = 0; m < M; ++m) {
            const FxtJob& j = hj[(size_t)m];
            a.take<float>((size_t)j.net.P); a.take<float>((size_t)j.net.P); a.take<float>((size_t)j.net.P);
            a.take<float>((size_t)j.total_steps);
        }
        a.take<char>(0); bytes_a = a.off;
        a.take<FxtJob>((size_t)M); a.take<uint8_t>(256); a.take<uint8_t>((size_t)n * L); a.take<float>((size_t)n);
        for (int m = 0; m < M; ++m) {
            const FxtJob& j = hj[(size_t)m];
            a.take<int32_t>((size_t)j.total_steps * j.batch);
            a.take<float>((size_t)j.total_steps);
            if (jobs[m].keep) a.take<uint8_t>((size_t)j.total_steps * j.batch * j.net.H);
        }
        a.take<char>(0); bytes_ab = a.off;
        a.take<FxtBar>(1);
        for (int m = 0; m < M; ++m) {
            const FxtJob& j = hj[(size_t)m];
            a.take<float>((size_t)j.S * (size_t)j.pstride);
            a.take<float>((size_t)j.S * (size_t)j.ws_slice);
        }
    };
    { Arena probe; plan(probe); need = probe.off + 256; }
    if (need > e->train_bytes) {
        if (e->d_train) {
            FX_HIP(e, hipStreamSynchronize(e->stream));
            FX_HIP(e, hipFree(e->d_train));
            e->d_train = nullptr; e->train_bytes = 0;
        }
        const size_t cap = need + need / 4;
        if (hipMalloc(&e->d_train, cap) != hipSuccess) { (void)hipGetLastError(); return fx_fail(e, FX_ENOMEM, "hipMalloc of the training arena failed"); }
        e->train_bytes = cap;
    }
    if (bytes_ab > e->train_host_bytes) {
        if (e->h_train) {
            FX_HIP(e, hipStreamSynchronize(e->stream));
            FX_HIP(e, hipHostFree(e->h_train));
            e->h_train = nullptr; e->train_host_bytes = 0;
        }
        const size_t cap = bytes_ab + bytes_ab / 4;
        if (hipHostMalloc(&e->h_train, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fx_fail(e, FX_ENOMEM, "hipHostMalloc of the training staging image failed"); }
        e->train_host_bytes = cap;
    }
    Arena a; a.base = (char*)e->d_train;
    hipStream_t st = e->stream;
    // the host image's address of a device address in A or B
    auto image = [&](const void* d) { return (char*)e->h_train + ((const char*)d - (const char*)e->d_train); };
    unsigned h_abort = 0;
    for (int m = 0; m < M; ++m) {                                  // region A
        FxtJob& j = hj[(size_t)m];
        const size_t P = (size_t)j.net.P;
        j.w = a.take<float>(P); j.adam_m = a.take<float>(P); j.adam_v = a.take<float>(P);
        j.step_loss = a.take<float>((size_t)j.total_steps);
        std::memcpy(image(j.w), jobs[m].weights, sizeof(float) * P);
        std::memcpy(image(j.adam_m), jobs[m].adam_m, sizeof(float) * P);
        std::memcpy(image(j.adam_v), jobs[m].adam_v, sizeof(float) * P);
    }
    a.take<char>(0);
    FxtJob* d_jobs = a.take<FxtJob>((size_t)M);                    // region B
    uint8_t* d_lut = a.take<uint8_t>(256);
    uint8_t* d_ascii = a.take<uint8_t>((size_t)n * L);
    float* d_labels = a.take<float>((size_t)n);
    std::memcpy(image(d_lut), lut, 256);
    std::memcpy(image(d_ascii), ascii, (size_t)n * L);
    std::memcpy(image(d_labels), labels, sizeof(float) * (size_t)n);
    for (int m = 0; m < M; ++m) {
        FxtJob& j = hj[(size_t)m];
        int32_t* d_order = a.take<int32_t>((size_t)j.total_steps * j.batch);
        j.order = d_order;
        float* d_lr = a.take<float>((size_t)j.total_steps);
        j.lr_t = d_lr;
        std::memcpy(image(d_order), jobs[m].order, sizeof(int32_t) * (size_t)j.total_steps * j.batch);
        std::memcpy(image(d_lr), lr[(size_t)m].data(), sizeof(float) * (size_t)j.total_steps);
        if (jobs[m].keep) {
            uint8_t* d_keep = a.take<uint8_t>((size_t)j.total_steps * j.batch * j.net.H);
            j.keep = d_keep;
            std::memcpy(image(d_keep), jobs[m].keep, (size_t)j.total_steps * j.batch * j.net.H);
        }
    }
    a.take<char>(0);
    FxtBar* d_bar = a.take<FxtBar>(1);                             // region C
    for (int m = 0; m < M; ++m) {
        FxtJob& j = hj[(size_t)m];
        j.partial = a.take<float>((size_t)j.S * (size_t)j.pstride);
        j.ws = a.take<float>((size_t)j.S * (size_t)j.ws_slice);
    }
    if (e->train_trace) {
        if (!e->d_train_dbg && hipMalloc(reinterpret_cast<void**>(&e->d_train_dbg), 64 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            return fx_fail(e, FX_ENOMEM, "hipMalloc of the training trace failed");
        }
        FX_HIP(e, hipMemsetAsync(e->d_train_dbg, 0, 64 * sizeof(unsigned long long), st));
        hj[0].dbg = e->d_train_dbg;
    }
    // one launch for the whole fit when every workgroup finds a CU at once (the step barriers need them co-resident)
    bool persistent = e->train_persistent != 0 && M <= 64 && !e->train_trace && !any_swz;     // (the one-launch fit has no rotated-row form)
    if (persistent) {
        if (lds_bytes > 48 * 1024) {
            static bool attr_fit[64] = {};
            if (!attr_fit[e->device & 63]) {
                FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_fit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_BUDGET));
                attr_fit[e->device & 63] = true;
            }
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_train_fit), threads, lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            per_cu = 0;
        }
        // (a margin of a few CUs: a kernel of another stream or process may hold one for a while)
        persistent = per_cu >= 1 && (int64_t)max_S * M <= (int64_t)per_cu * (e->num_cus - 8);
    }
    for (FxtJob& j : hj) j.agent_io = persistent ? 1 : 0;
    std::memcpy(image(d_jobs), hj.data(), sizeof(FxtJob) * (size_t)M);
    e->train_prof_ns[0] = since();
    FX_HIP(e, hipMemcpyAsync(e->d_train, e->h_train, bytes_ab, hipMemcpyHostToDevice, st));      // regions A + B, one copy
    e->train_prof_ns[1] = since();

    const dim3 grid_fb((unsigned)max_S, (unsigned)M), grid_adam((unsigned)(((max_P + 3) / 4 + 63) / 64), (unsigned)M);      // four parameters per thread (fxt_adam4), one wave per workgroup: a GFP-length CNN's 10 344 quads spread over 162 workgroups per member
    if (lds_bytes > 48 * 1024) {
        static bool attr_set[64] = {};
        if (!attr_set[e->device & 63]) {
            FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_fb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_BUDGET));
            attr_set[e->device & 63] = true;
        }
        static bool attr_swz[64] = {};
        if (any_swz && !attr_swz[e->device & 63]) {
            FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_fb_swz), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_BUDGET));
            FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_fb_c32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_BUDGET));
            FX_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_fb_c32p), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_BUDGET));
            attr_swz[e->device & 63] = true;
        }
    }
    if (persistent) {
        FX_HIP(e, hipMemsetAsync(d_bar, 0, sizeof(FxtBar), st));
        hipLaunchKernelGGL(k_train_fit, grid_fb, dim3((unsigned)threads), lds_bytes, st, d_jobs, d_ascii, d_lut, d_labels, d_bar);
        FX_HIP(e, hipGetLastError());
        FX_HIP(e, hipMemcpyAsync(&h_abort, &d_bar->abort, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    } else {
        for (int s = 0; s < max_steps; ++s) {
            if (c32p) hipLaunchKernelGGL(k_train_fb_c32p, grid_fb, dim3((unsigned)threads), lds_bytes, st, d_jobs, s, d_ascii, d_lut, d_labels);
            else if (c32) hipLaunchKernelGGL(k_train_fb_c32, grid_fb, dim3((unsigned)threads), lds_bytes, st, d_jobs, s, d_ascii, d_lut, d_labels);
            else if (any_swz) hipLaunchKernelGGL(k_train_fb_swz, grid_fb, dim3((unsigned)threads), lds_bytes, st, d_jobs, s, d_ascii, d_lut, d_labels);
            else hipLaunchKernelGGL(k_train_fb, grid_fb, dim3((unsigned)threads), lds_bytes, st, d_jobs, s, d_ascii, d_lut, d_labels);
            hipLaunchKernelGGL(k_train_adam, grid_adam, dim3(64), 0, st, d_jobs, s);
        }
        FX_HIP(e, hipGetLastError());
    }
    e->train_prof_ns[2] = since();
    FX_HIP(e, hipMemcpyAsync(e->h_train, e->d_train, bytes_a, hipMemcpyDeviceToHost, st));         // region A, one copy
    FX_HIP(e, hipStreamSynchronize(st));
    e->train_prof_ns[3] = since();
    if (h_abort) return fx_fail(e, FX_ESTATE, "fx_train_fit: a step barrier of the one-launch fit was not passed within 2 s (workgroups not co-resident?); "
                                              "set the engine option train_persistent = 0 for a launch per step");
    for (int m = 0; m < M; ++m) {
        const FxtJob& j = hj[(size_t)m];
        const size_t P = (size_t)j.net.P;
        std::memcpy(jobs[m].weights, image(j.w), sizeof(float) * P);
        std::memcpy(jobs[m].adam_m, image(j.adam_m), sizeof(float) * P);
        std::memcpy(jobs[m].adam_v, image(j.adam_v), sizeof(float) * P);
        if (jobs[m].step_loss && j.total_steps > 0) std::memcpy(jobs[m].step_loss, image(j.step_loss), sizeof(float) * (size_t)j.total_steps);
        jobs[m].step += hj[(size_t)m].total_steps;
        e->counters.train_steps += hj[(size_t)m].total_steps;
    }
    e->train_prof_ns[4] = since();
    return FX_OK;
}

// Epoch shuffles of one fit: see include/flexs_amd.h.
int fx_train_orders(uint64_t seed, int64_t n, int epochs, int32_t* out) {
    if (n < 0 || n > ((int64_t)1 << 30) || epochs < 0 || (!out && n > 0 && epochs > 0)) return FX_EINVAL;
    uint64_t s[4];
    for (int i = 0; i < 4; ++i) {                          // splitmix64: four state words that are never all zero
        uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        s[i] = z ^ (z >> 31);
    }
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    auto next = [&]() {                                    // xoshiro256**
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    };
    auto below = [&](uint64_t bound) {                     // uniform in [0, bound): multiply-high with rejection (Lemire)
        unsigned __int128 m = (unsigned __int128)next() * bound;
        uint64_t lo = (uint64_t)m;
        if (lo < bound) {
            const uint64_t floor = (0 - bound) % bound;
            while (lo < floor) { m = (unsigned __int128)next() * bound; lo = (uint64_t)m; }
        }
        return (uint64_t)(m >> 64);
    };
    for (int e_ = 0; e_ < epochs; ++e_) {
        int32_t* p = out + (int64_t)e_ * n;
        for (int64_t i = 0; i < n; ++i) p[i] = (int32_t)i;
        for (int64_t i = n - 1; i > 0; --i) {
            const int64_t j = (int64_t)below((uint64_t)i + 1);
            const int32_t t = p[i]; p[i] = p[j]; p[j] = t;
        }
    }
    return FX_OK;
}

// Test hook, host only (no GPU): ONE mini-batch step of one member through the HOST build of train_core.h -- the
// same source the kernels are compiled from, threads as loops, the MFMA as an fmaf chain.  tests/test_train_native.py
// holds it to oracle/train_np.py on the CPU.  rows = the mini-batch (<= 4096 rows), slices of `R` rows.
int fx_debug_train_step_host(int kind, int L, int A, int F, int H, int K, float* weights, float* adam_m, float* adam_v,
                             int64_t* step, const uint8_t* ascii, int rows, const uint8_t lut[256], const float* labels,
                             const uint8_t* keep, int R, float* loss_out) {
    if (!weights || !adam_m || !adam_v || !step || !ascii || !lut || !labels || rows < 1 || rows > 4096 || R < 1 || R > 64) return FX_EINVAL;
    if (kind < FX_CNN || kind > FX_GE) return FX_EINVAL;
    FxtJob j{};
    j.net = fxt_net(kind, L, A, kind == FX_CNN ? F : 0, H, kind == FX_CNN ? K : 0);
    j.batch = rows; j.steps_per_epoch = 1; j.total_steps = 1; j.n = rows;
    j.R = R; j.S = (rows + R - 1) / R;
    j.w = weights; j.adam_m = adam_m; j.adam_v = adam_v;
    std::vector<float> partial((size_t)j.S * (j.net.P + 1), 0.f);
    j.partial = partial.data();
    std::vector<int32_t> order((size_t)rows);
    for (int i = 0; i < rows; ++i) order[(size_t)i] = i;
    j.order = order.data();
    j.keep = keep;
    j.seed = 0;
    const double t = (double)(*step + 1);
    const float lr_t = (float)(FXT_LR * std::sqrt(1.0 - std::pow(FXT_BETA_2, t)) / (1.0 - std::pow(FXT_BETA_1, t)));
    j.lr_t = &lr_t;
    j.ws_slice = fxt_ws(j.net, R).total;
    std::vector<float> ws((size_t)j.S * (size_t)j.ws_slice, 0.f);
    j.ws = ws.data();
    float loss = 0.f;
    j.step_loss = &loss;
    for (int s = 0; s < j.S; ++s) fxt_forward_backward<0, 0>(j, FxtWg{0, 1}, 0, s, ascii, lut, labels, j.ws + (long long)s * j.ws_slice, (const float*)j.w);
    fxt_step_loss(j, 0);
    for (int i = 0; i < j.net.P; ++i) fxt_adam(j, 0, i);
    *step += 1;
    if (loss_out) *loss_out = loss;
    return FX_OK;
}
