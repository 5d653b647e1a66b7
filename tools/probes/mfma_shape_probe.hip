// v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32 on gfx950 under the instruction mix of the scoring kernels
// (round-3 verdict, Next #3): per 4 MFMAs one ds_read_b128 of A operands from LDS, V VALU instructions per 2048 FLOP,
// NACC independent accumulators per wave (the kernels chain 40-160 dependent MFMAs into each), W waves per SIMD.
// Both shapes do 64 FLOP per cycle per SIMD (8 passes x 2048 FLOP vs 16 passes x 4096 FLOP); what can differ is how
// well issue slots, LDS reads and dependent chains hide.  Reported: ns per 2048 FLOP per SIMD (13.3 ns = the pipe at 2.4 GHz).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shape_probe.hip -o /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 16: NACC f4 accumulators; per iteration: one ds_read_b128 (4 A values) then 4 k-steps x NACC MFMAs
template <int NACC, int V, bool LDS>
__global__ void __launch_bounds__(1024) k16(float* out, int iters) {
    __shared__ f4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (f4){1e-3f * i, 1e-4f, 1e-5f, 1e-6f};
    __syncthreads();
    f4 acc[NACC];
    float x[8];
    float b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)j;
    int idx = threadIdx.x & 63;
    f4 a = lds[idx];
    for (int i = 0; i < iters; ++i) {
        if (LDS) { idx = (idx + 64) & 4095; a = lds[idx]; asm volatile("" ::: "memory"); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b, acc[j], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[(j + v + r) & 7]) : "v"(b), "v"(b));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// SHAPE 32: NACC 16-register accumulators; per iteration one ds_read_b128 then 4 k-steps x NACC MFMAs (4096 FLOP each);
// V2 = VALU instructions per MFMA (= 2 V of the 16 shape for the same VALU work per FLOP)
template <int NACC, int V2, bool LDS>
__global__ void __launch_bounds__(1024) k32(float* out, int iters) {
    __shared__ f4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (f4){1e-3f * i, 1e-4f, 1e-5f, 1e-6f};
    __syncthreads();
    f16v acc[NACC];
    float x[8];
    float b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)j;
    int idx = threadIdx.x & 63;
    f4 a = lds[idx];
    for (int i = 0; i < iters; ++i) {
        if (LDS) { idx = (idx + 64) & 4095; a = lds[idx]; asm volatile("" ::: "memory"); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b, acc[j], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V2; ++v) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[(j + v + r) & 7]) : "v"(b), "v"(b));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += acc[j][k];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static float time_kernel(K kern, int waves_per_simd, int iters, float* d_out) {
    const int threads = waves_per_simd * 4 * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d_out, iters / 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    float* d_out = nullptr;
    hipMalloc(&d_out, sizeof(float) * 256 * 1024);
    const int iters = 6000;
    printf("shape  nacc  valu/2048F  lds  waves/simd   ns per 2048 FLOP per SIMD\n");
#define ROW16(NACC, V, LDSF)                                                                                           \
    for (int w : {1, 2, 4}) {                                                                                          \
        const float ms = time_kernel(k16<NACC, V, LDSF>, w, iters, d_out);                                             \
        printf("16x16x4  %d  %d  %d  %d  %.2f\n", NACC, V, (int)LDSF, w, ms * 1e6 / ((double)iters * 4 * NACC * w));  \
    }
#define ROW32(NACC, V2, LDSF)                                                                                          \
    for (int w : {1, 2, 4}) {                                                                                          \
        const float ms = time_kernel(k32<NACC, V2, LDSF>, w, iters, d_out);                                            \
        printf("32x32x2  %d  %d  %d  %d  %.2f\n", NACC, V2 / 2, (int)LDSF, w, ms * 1e6 / ((double)iters * 4 * NACC * w * 2)); \
    }
    // the headline CNN kernel: 2 accumulators per wave (FT = 2, one tile), ~0.6 VALU per MFMA; MLP: 7 accumulators, ~1.9
    ROW16(2, 0, false) ROW16(2, 0, true) ROW16(2, 1, true) ROW16(2, 2, true) ROW16(8, 1, true) ROW16(8, 2, true)
    ROW32(1, 0, false) ROW32(1, 0, true) ROW32(1, 2, true) ROW32(1, 4, true) ROW32(2, 0, true) ROW32(2, 2, true) ROW32(2, 4, true) ROW32(4, 2, true) ROW32(4, 4, true)
    fflush(stdout);
    return 0;
}
