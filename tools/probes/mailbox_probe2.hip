// Second mailbox probe (gfx950): (A) can the host store straight into device memory (large BAR)?  (B) round trip of a
// request answered by G resident workgroups, for G = 1..128, with the request word (1) in mapped host memory, polled over
// PCIe, or (2) in device memory, written by the host through the BAR and polled locally; answers are 16 floats per
// workgroup in host memory, detected by the host through a sentinel (no flag, no ordering assumption between lines).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <sys/wait.h>
#include <unistd.h>
#include <immintrin.h>

#define CK(x) do { hipError_t rc_ = (x); if (rc_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(rc_)); fflush(stdout); } } while (0)

struct Ctl { alignas(64) volatile uint64_t req; alignas(64) volatile uint32_t stop; };

__global__ void k_server(volatile Ctl* ctl, float* answers, unsigned long long idle_ticks, unsigned long long life_ticks, int mode) {
    __shared__ unsigned long long s_req;
    __shared__ int s_exit;
    unsigned long long last = 0;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_last = t_start;
    for (;;) {
        if (threadIdx.x == 0) {
            int ex = 0;
            unsigned long long r;
            for (;;) {
                r = __hip_atomic_load(const_cast<uint64_t*>(&ctl->req), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (r != last) break;
                const unsigned long long now = wall_clock64();
                if (__hip_atomic_load(const_cast<uint32_t*>(&ctl->stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || now - t_last > idle_ticks ||
                    now - t_start > life_ticks) { ex = 1; break; }
            }
            s_req = r; s_exit = ex;
        }
        __syncthreads();
        if (s_exit) break;
        const unsigned long long r = s_req;
        if (mode & 2) {                                            // (score, tag) pairs: one 8-byte store per sequence
            if (threadIdx.x < 16) {
                float2 v; v.x = (float)(r & 0xFFFF) + threadIdx.x; v.y = __uint_as_float((unsigned)(r >> 16));
                reinterpret_cast<float2*>(answers)[blockIdx.x * 16 + threadIdx.x] = v;
            }
        } else {
            if (threadIdx.x < 16) answers[blockIdx.x * 16 + threadIdx.x] = (float)(r & 0xFFFF) + threadIdx.x;
        }
        if (mode & 1) __threadfence_system();
        if (threadIdx.x == 0) t_last = wall_clock64();
        last = r;
        __syncthreads();
    }
}

static int child_write_test(int kind) {
    pid_t pid = fork();
    if (pid == 0) {
        void* p = nullptr;
        hipError_t rc = kind == 0 ? hipMalloc(&p, 4096) : hipExtMallocWithFlags(&p, 4096, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (rc != hipSuccess) _exit(3);
        *(volatile uint32_t*)p = 0x1234u;                          // SIGSEGV if the allocation is not mapped for the host
        uint32_t back = 0;
        hipMemcpy(&back, p, 4, hipMemcpyDeviceToHost);
        _exit(back == 0x1234u ? 0 : 4);
    }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) return -WTERMSIG(st);
    return WEXITSTATUS(st);
}

int main(int argc, char** argv) {
    // (A) before any HIP call in this process (children initialise their own runtime)
    int okk[3];
    for (int k = 0; k < 3; ++k) {
        okk[k] = argc > 1 ? atoi(argv[1]) : child_write_test(k);
        printf("host store into %s: %s (%d)\n", k == 0 ? "hipMalloc memory" : k == 1 ? "fine-grained device memory" : "uncached device memory",
               okk[k] == 0 ? "works, value read back" : okk[k] < 0 ? "signal" : "failed", okk[k]);
    }
    fflush(stdout);
    float* h_ans = nullptr; float* d_ans = nullptr;
    CK(hipHostMalloc(&h_ans, 128 * 16 * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_ans, h_ans, 0));
    Ctl* h_ctl = nullptr; Ctl* d_ctl_host = nullptr;
    CK(hipHostMalloc(&h_ctl, sizeof(Ctl), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_ctl_host, h_ctl, 0));
    Ctl* dev_ctl = nullptr;
    int dev_kind = argc > 2 ? atoi(argv[2]) : okk[1] == 0 ? 1 : okk[2] == 0 ? 2 : okk[0] == 0 ? 0 : -1;
    printf("device-memory request word: kind %d\n", dev_kind);
    if (dev_kind == 0) CK(hipMalloc(&dev_ctl, sizeof(Ctl)));
    if (dev_kind > 0) CK(hipExtMallocWithFlags((void**)&dev_ctl, sizeof(Ctl), dev_kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const uint32_t SENT = 0x7FC0DEADu;
    for (int mode = 1; mode < 4; mode += 2)
    for (int scheme = 0; scheme < 2; ++scheme) {
        if (scheme == 1 && !dev_ctl) { printf("no host-writable device memory: scheme 2 skipped\n"); break; }
        volatile Ctl* host_view = scheme == 0 ? h_ctl : dev_ctl;      // where the HOST writes
        Ctl* dev_view = scheme == 0 ? d_ctl_host : dev_ctl;           // where the DEVICE polls
        bool ok_prev = true;
        for (int G : {1, 3, 16, 48, 96}) {
            if (!ok_prev && G > 1) continue;
            host_view->req = 0; host_view->stop = 0;
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_server, dim3(G), dim3(256), 0, s1, dev_view, d_ans, 200000ull, 20000000ull, mode);
            CK(hipGetLastError());
            std::vector<double> ts;
            bool ok = true;
            for (int it = 0; it < 3000 && ok; ++it) {
                if (!(mode & 2)) for (int i = 0; i < G * 16; ++i) ((volatile uint32_t*)h_ans)[i] = SENT;
                auto a = std::chrono::steady_clock::now();
                __atomic_thread_fence(__ATOMIC_RELEASE);
                host_view->req = ((uint64_t)(it + 1) << 16) | 7u;
                _mm_sfence();                                        // (the BAR is write-combining on the host side: push the store out)
                for (int i = 0; i < G * 16; ++i) {
                    unsigned spins = 0;
                    while ((mode & 2) ? ((volatile uint32_t*)h_ans)[2 * i + 1] != (uint32_t)(it + 1) : ((volatile uint32_t*)h_ans)[i] == SENT) {
                        if ((++spins & 0xFFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 0.5) { ok = false; break; }
                    }
                    if (!ok) break;
                }
                auto b = std::chrono::steady_clock::now();
                if (it >= 200) ts.push_back(std::chrono::duration<double>(b - a).count() * 1e6);
                if (ok && h_ans[(mode & 2) ? 10 : 5] != 7.f + 5.f) { printf("wrong answer %f\n", h_ans[5]); ok = false; }
            }
            host_view->stop = 1;
            CK(hipStreamSynchronize(s1));
            if (!ok || ts.empty()) {
                printf("mode %d scheme %d G=%d: timed out after %zu timed requests; answers[0..3] = %08x %08x %08x %08x, req %llx\n", mode, scheme + 1, G, ts.size(),
                       ((volatile uint32_t*)h_ans)[0], ((volatile uint32_t*)h_ans)[1], ((volatile uint32_t*)h_ans)[2], ((volatile uint32_t*)h_ans)[3],
                       (unsigned long long)host_view->req);
                ok_prev = false;
                continue;
            }
            std::sort(ts.begin(), ts.end());
            printf("[%s, %s] request word in %s, %3d resident workgroups: round trip median %.2f us, p90 %.2f us, p99 %.2f us\n",
                   (mode & 2) ? "tagged answers" : "sentinel answers", (mode & 1) ? "system fence" : "no fence",
                   scheme == 0 ? "host memory (polled over PCIe)" : "device memory (host writes through the BAR)", G,
                   ts[ts.size() / 2], ts[ts.size() * 9 / 10], ts[ts.size() * 99 / 100]);
            fflush(stdout);
        }
    }
    return 0;
}
