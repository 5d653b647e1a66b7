// Round-trip latency of a host <-> persistent-kernel mailbox in mapped pinned memory (gfx950): the host posts a sequence
// number, a resident workgroup polls it with system-scope loads, does `work_us` of busy work, writes a result + done flag;
// the host spins on the flag.  Compared with: empty-kernel launch + hipStreamSynchronize.  The kernel exits by itself after
// `idle_ms` without requests or after `life_ms` (no way to hang the device).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct Mail {
    alignas(64) volatile uint64_t req;
    alignas(64) volatile uint32_t stop;
    alignas(64) volatile uint64_t done;
    alignas(64) volatile uint32_t alive;
    alignas(64) float payload[64];
    alignas(64) float result[64];
};

__global__ void k_server(Mail* m, int work_iters, unsigned long long idle_ticks, unsigned long long life_ticks) {
    __shared__ unsigned long long s_req;
    __shared__ int s_exit;
    unsigned long long last = 0;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_last = t_start;
    if (threadIdx.x == 0) __hip_atomic_store(&m->alive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        if (threadIdx.x == 0) {
            int ex = 0;
            unsigned long long r;
            for (;;) {
                r = __hip_atomic_load(&m->req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (r != last) break;
                const unsigned long long now = wall_clock64();
                if (__hip_atomic_load(&m->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || now - t_last > idle_ticks || now - t_start > life_ticks) { ex = 1; break; }
            }
            s_req = r; s_exit = ex;
        }
        __syncthreads();
        if (s_exit) break;
        const unsigned long long r = s_req;
        float v = m->payload[threadIdx.x & 63];
        for (int i = 0; i < work_iters; ++i) v = v * 1.0000001f + 1e-7f;
        if (threadIdx.x < 64) m->result[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(&m->done, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            t_last = wall_clock64();
        }
        last = r;
        __syncthreads();
    }
    if (threadIdx.x == 0) __hip_atomic_store(&m->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_empty(float* p) { if (threadIdx.x == 0 && p) p[0] = 1.f; }

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    Mail* h = nullptr; Mail* d = nullptr;
    if (hipHostMalloc(&h, sizeof(Mail), hipHostMallocMapped) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    new (h) Mail();
    h->req = 0; h->stop = 0; h->done = 0; h->alive = 0;
    hipHostGetDevicePointer((void**)&d, h, 0);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int work : {0, 2000}) {
        h->req = 0; h->done = 0; h->stop = 0;
        hipLaunchKernelGGL(k_server, dim3(3), dim3(768), 0, s1, d, work, 200000ull /* 2 ms idle */, 5000000ull /* 50 ms life */);
        auto t0 = std::chrono::steady_clock::now();
        while (!h->alive) { if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { printf("server did not start\n"); return 1; } }
        std::vector<double> ts;
        uint64_t seq = 0;
        bool ok = true;
        for (int it = 0; it < 2000 && ok; ++it) {
            for (int i = 0; i < 64; ++i) h->payload[i] = (float)i;
            auto a = std::chrono::steady_clock::now();
            std::atomic_thread_fence(std::memory_order_release);
            h->req = ++seq;
            while (h->done != seq) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 0.01) { printf("timeout at %d\n", it); ok = false; break; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() * 1e6);
        }
        h->stop = 1;
        hipStreamSynchronize(s1);
        if (!ts.empty()) printf("mailbox round trip, %d busy iterations per request (3 workgroups x 768 threads; flag seen from workgroup 0): median %.2f us, min %.2f us, p99 %.2f us\n",
                                work, med(ts), ts.front(), ts[(size_t)(ts.size() * 0.99)]);
    }
    float* dp = nullptr; hipMalloc(&dp, 64);
    std::vector<double> ts;
    for (int it = 0; it < 2000; ++it) {
        auto a = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_empty, dim3(3), dim3(768), 0, s2, dp);
        hipStreamSynchronize(s2);
        ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() * 1e6);
    }
    printf("empty kernel launch + hipStreamSynchronize: median %.2f us, min %.2f us\n", med(ts), ts.front());
    ts.clear();
    for (int it = 0; it < 2000; ++it) {
        auto a = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_empty, dim3(3), dim3(768), 0, s2, dp);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s2, dp);
        hipStreamSynchronize(s2);
        ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() * 1e6);
    }
    printf("two empty kernels + hipStreamSynchronize: median %.2f us, min %.2f us\n", med(ts), ts.front());
    return 0;
}
