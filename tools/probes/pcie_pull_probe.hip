// How fast can workgroups pull pinned host memory over PCIe (zero-copy reads), by grid size and loads in flight per lane?
// The relay of a launched-first ensemble call (fx_common.h FxRelay) moves 9 MB in ~300 us = ~30 GB/s from 32 workgroups:
// is that the platform's rate for kernel-issued reads, or the relay's own shape?   hipcc --offload-arch=gfx950 -O3 -o pcie_pull_probe pcie_pull_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
struct __attribute__((packed, aligned(1))) B16 { unsigned w[4]; };
template <int INFLIGHT>
__global__ void __launch_bounds__(1024) k_pull(const unsigned char* src, size_t bytes, unsigned* sink, int chunk) {
    // every wave walks chunks of `chunk` bytes (a tile: 1440 bytes for 16 x 90 residues), lanes take 16 bytes each
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t n_chunks = bytes / chunk;
    unsigned acc = 0;
    for (size_t c0 = wave * INFLIGHT; c0 < n_chunks; c0 += waves * INFLIGHT) {
        B16 v[INFLIGHT][2];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int off = (lane + 64 * h) * 16;
                v[k][h] = B16{{0, 0, 0, 0}};
                if (c0 + k < n_chunks && off + 16 <= chunk) v[k][h] = *reinterpret_cast<const B16*>(src + (c0 + k) * chunk + off);
            }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc += v[k][h].w[0] ^ v[k][h].w[1] ^ v[k][h].w[2] ^ v[k][h].w[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const size_t bytes = 36u << 20;
    unsigned char* h = nullptr; unsigned* sink = nullptr; unsigned char* d = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&h), bytes, hipHostMallocMapped));
    for (size_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 2654435761u >> 24);
    unsigned char* hd = nullptr; CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64)); CK(hipMalloc(reinterpret_cast<void**>(&d), bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_it = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e9f; for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; } return best; };
    { const float ms = time_it([&] { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0)); }); std::printf("hipMemcpyAsync H2D of %zu MB: %.1f us = %.1f GB/s\n", bytes >> 20, ms * 1e3, bytes / ms / 1e6); }
    for (int chunk : {1440, 4096}) for (int grid : {32, 64, 128, 256}) for (int waves : {4, 16}) {
        const float t1 = time_it([&] { hipLaunchKernelGGL(k_pull<1>, dim3(grid), dim3(waves * 64), 0, 0, hd, bytes, sink, chunk); });
        const float t4 = time_it([&] { hipLaunchKernelGGL(k_pull<4>, dim3(grid), dim3(waves * 64), 0, 0, hd, bytes, sink, chunk); });
        std::printf("chunks of %d bytes, %3d workgroups x %2d waves: 1 chunk in flight per wave %.1f GB/s, 4 in flight %.1f GB/s\n", chunk, grid, waves, bytes / t1 / 1e6, bytes / t4 / 1e6);
    }
    return 0;
}
