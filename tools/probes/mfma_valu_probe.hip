// Does VALU work issued between v_mfma_f32_16x16x4_f32 instructions steal matrix-pipe time on gfx950?
// Each wave runs ITER x 8 MFMAs on 8 independent accumulators with V independent v_fmac_f32 after every MFMA.
// Reported: time per MFMA per SIMD in ns for W waves per SIMD (W = 1, 2, 4) and V = 0..6.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o tools/probes/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ void __launch_bounds__(1024) k_probe(float* out, int iters) {
    f4 acc[8];
    float x[8];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = (f4){0.f, 0.f, 0.f, 0.f}; x[j] = (float)j; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[(j + v) & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w + x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
static float run(int waves_per_simd, int iters, float* d_out) {
    const int threads = waves_per_simd * 4 * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<V>, dim3(256), dim3(threads), 0, 0, d_out, iters / 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<V>, dim3(256), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d_out = nullptr;
    hipMalloc(&d_out, sizeof(float) * 256 * 1024);
    const int iters = 20000;
    printf("{\"probe\": \"mfma_valu\", \"rows\": [\n");
    bool first = true;
    for (int w : {1, 2, 4}) {
        float ms[7];
        ms[0] = run<0>(w, iters, d_out); ms[1] = run<1>(w, iters, d_out); ms[2] = run<2>(w, iters, d_out);
        ms[3] = run<3>(w, iters, d_out); ms[4] = run<4>(w, iters, d_out); ms[5] = run<5>(w, iters, d_out);
        ms[6] = run<6>(w, iters, d_out);
        for (int v = 0; v < 7; ++v) {
            const double mfma_per_simd = (double)iters * 8 * w;                 // MFMAs one SIMD executes
            printf("%s {\"waves_per_simd\": %d, \"valu_per_mfma\": %d, \"ms\": %.4f, \"ns_per_mfma_per_simd\": %.3f}",
                   first ? "" : ",\n", w, v, ms[v], ms[v] * 1e6 / mfma_per_simd);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
