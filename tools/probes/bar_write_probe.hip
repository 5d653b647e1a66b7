// How fast does one host core store into device memory through the BAR?  (The resident form's request bytes go that way.)
// Fine-grained device memory, host pointer = device pointer (large BAR).  Per size and copy routine: median time of
// copy + sfence + read-back of the last word (the read-back cannot pass the posted writes: it returns when they have landed),
// minus the time of the read-back alone.
//   hipcc -O2 --offload-arch=gfx950 bar_write_probe.hip -o bar_write_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void copy_memcpy(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
__attribute__((target("avx2"))) static void copy_stream256(void* d, const void* s, size_t n) {
    for (size_t i = 0; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)((char*)d + i), _mm256_loadu_si256((const __m256i*)((const char*)s + i)));
}
__attribute__((target("avx2"))) static void copy_store256(void* d, const void* s, size_t n) {
    for (size_t i = 0; i + 32 <= n; i += 32) _mm256_store_si256((__m256i*)((char*)d + i), _mm256_loadu_si256((const __m256i*)((const char*)s + i)));
}
__attribute__((target("avx512f"))) static void copy_stream512(void* d, const void* s, size_t n) {
    for (size_t i = 0; i + 64 <= n; i += 64) _mm512_stream_si512((__m512i*)((char*)d + i), _mm512_loadu_si512((const void*)((const char*)s + i)));
}
static void copy_movsb(void* d, const void* s, size_t n) { asm volatile("rep movsb" : "+D"(d), "+S"(s), "+c"(n) :: "memory"); }
static void copy_u64(void* d, const void* s, size_t n) {
    for (size_t i = 0; i + 8 <= n; i += 8) { unsigned long long v; std::memcpy(&v, (const char*)s + i, 8); *(volatile unsigned long long*)((char*)d + i) = v; }
}

int main() {
    CK(hipSetDevice(0));
    char* dev = nullptr;
    const size_t cap = 1 << 20;
    CK(hipExtMallocWithFlags((void**)&dev, cap, hipDeviceMallocFinegrained));
    std::vector<char> src(cap, 7);
    struct R { const char* name; void (*fn)(void*, const void*, size_t); bool ok; };
    R routines[] = {{"memcpy", copy_memcpy, true}, {"u64 stores", copy_u64, true}, {"rep movsb", copy_movsb, true},
                    {"avx2 store", copy_store256, (bool)__builtin_cpu_supports("avx2")}, {"avx2 stream", copy_stream256, (bool)__builtin_cpu_supports("avx2")},
                    {"avx512 stream", copy_stream512, (bool)__builtin_cpu_supports("avx512f")}};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> base;
    volatile unsigned sink = 0;
    for (int i = 0; i < 300; ++i) { auto t0 = now(); sink += *(volatile unsigned*)(dev + 64); base.push_back(std::chrono::duration<double>(now() - t0).count() * 1e6); }
    const double rb = med(base);
    printf("read-back of one word through the BAR: %.2f us\n", rb);
    printf("%-14s", "bytes");
    for (auto& r : routines) if (r.ok) printf("%16s", r.name);
    printf("   (us for the copy to land; GB/s)\n");
    for (size_t n : {1024, 4096, 8192, 16384, 32768, 65536, 262144}) {
        printf("%-14zu", n);
        for (auto& r : routines) {
            if (!r.ok) continue;
            std::vector<double> ts;
            for (int i = 0; i < 200; ++i) {
                auto t0 = now();
                r.fn(dev, src.data(), n);
                _mm_sfence();
                sink += *(volatile unsigned*)(dev + n - 4);
                ts.push_back(std::chrono::duration<double>(now() - t0).count() * 1e6);
            }
            const double t = med(ts) - rb;
            printf("  %6.2f (%5.1f)", t, n / t * 1e-3);
        }
        printf("\n");
    }
    // and the other direction of the alternative: the DEVICE reading request bytes from pinned host memory is measured by
    // tools/probes/mailbox_probe*.hip (answers go that way already)
    return 0;
}
