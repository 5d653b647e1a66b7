// What does waiting for a launched kernel cost the host?  One kernel that spins for W us, then
//   (a) hipStreamSynchronize,
//   (b) hipStreamWriteValue32 into pinned host memory behind the kernel + the host polling that word,
//   (c) the kernel's last instruction stores a flag to pinned host memory (system scope) + the host polling it,
//   (d) hipEventRecord + polling hipEventQuery.
// Reported: median of (time from before the launch to "the host knows") minus W.
//   hipcc -O2 --offload-arch=gfx950 sync_latency_probe.hip -o sync_latency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_spin(unsigned long long ticks, volatile unsigned* flag, unsigned value) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(const_cast<unsigned*>(flag), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* h = nullptr; CK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    unsigned* d = nullptr; CK(hipHostGetDevicePointer((void**)&d, h, 0));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%8s %12s %12s %12s %12s   (us beyond the kernel's own W)\n", "W us", "streamSync", "writeValue", "kernel flag", "eventQuery");
    for (double W : {0.0, 5.0, 30.0, 100.0}) {
        const unsigned long long ticks = (unsigned long long)(W * 100);
        double res[4];
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<double> ts;
            unsigned seq = 0;
            for (int i = 0; i < 220; ++i) {
                ++seq;
                auto t0 = now();
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, mode == 2 ? d : nullptr, seq);
                if (mode == 0) { CK(hipStreamSynchronize(st)); }
                else if (mode == 1) { CK(hipStreamWriteValue32(st, d + 16, seq, 0)); while (((volatile unsigned*)h)[16] != seq) _mm_pause(); }
                else if (mode == 2) { while (((volatile unsigned*)h)[0] != seq) _mm_pause(); }
                else { CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause(); }
                const double t = std::chrono::duration<double>(now() - t0).count() * 1e6;
                if (i >= 20) ts.push_back(t);
                CK(hipStreamSynchronize(st));
            }
            res[mode] = med(ts) - W;
        }
        printf("%8.1f %12.2f %12.2f %12.2f %12.2f\n", W, res[0], res[1], res[2], res[3]);
    }
    return 0;
}
