// Host side of the resident form: the fast path of the answer collection (fx_resident.hip server_call).  Plain C++, no GPU code --
// kept out of the HIP translation units because it uses x86 vector intrinsics.
//
// An answer is one 8-byte word in pinned host memory, written by the device in one store: bits 0-31 the score (float32),
// bits 32-62 the request's tag, bit 63 "a character outside the alphabet".  The scalar loop of round 3 -- load, tag compare,
// store, per answer -- ran at ~1.1 ns per answer with every line already landed (profiles/r4_server_collect.log): 2.2 us per
// member of a 2001-sequence request, most of what such a call still cost after the first answer.  Here a 64-byte line (eight
// answers) is tested with two vector compares and its eight scores leave with one 32-byte store; the scalar loop only sees
// the lines whose answers have not all arrived yet (it waits for them one by one, as before).
#include <cstdint>
#include <immintrin.h>

extern "C" {

// Answers [n0, N) of one member, n0 a multiple of 8: whole lines whose eight tags all equal `seq` are unpacked into
// scores[n0 ...]; stops at the first line with an answer still missing (or fewer than 8 answers left) and returns its index.
// `*bad` is set when an answer carries the bad-character bit.  `ahead`: answers ahead of the walk to prefetch.
__attribute__((target("avx2")))
static int64_t collect_lines_avx2(const volatile unsigned long long* ans, float* scores, int64_t n0, int64_t N, unsigned seq,
                                  int64_t ahead, int* bad) {
    const __m256i tagmask = _mm256_set1_epi64x(0x7FFFFFFF00000000ll);
    const __m256i want = _mm256_set1_epi64x((long long)((unsigned long long)seq << 32));
    const __m256i pick = _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6);
    __m256i flags = _mm256_setzero_si256();
    int64_t n = n0;
    for (; n + 8 <= N; n += 8) {
        const unsigned long long* p = const_cast<const unsigned long long*>(ans) + n;
        __builtin_prefetch(p + ahead, 0, 0);
        const __m256i a0 = _mm256_load_si256(reinterpret_cast<const __m256i*>(p));
        const __m256i a1 = _mm256_load_si256(reinterpret_cast<const __m256i*>(p + 4));
        const __m256i ok = _mm256_and_si256(_mm256_cmpeq_epi64(_mm256_and_si256(a0, tagmask), want),
                                            _mm256_cmpeq_epi64(_mm256_and_si256(a1, tagmask), want));
        if (_mm256_movemask_epi8(ok) != -1) break;
        flags = _mm256_or_si256(flags, _mm256_or_si256(a0, a1));
        const __m128i lo = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(a0, pick));
        const __m128i hi = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(a1, pick));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(scores + n), lo);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(scores + n + 4), hi);
    }
    if (_mm256_movemask_pd(_mm256_castsi256_pd(flags))) *bad = 1;       // (bit 63 of any collected answer)
    return n;
}

int64_t fx_collect_lines(const volatile unsigned long long* ans, float* scores, int64_t n0, int64_t N, unsigned seq, int64_t ahead, int* bad) {
    static const int have_avx2 = __builtin_cpu_supports("avx2");
    if (!have_avx2 || (n0 & 7) || (reinterpret_cast<uintptr_t>(const_cast<const unsigned long long*>(ans)) & 63)) return n0;
    return collect_lines_avx2(ans, scores, n0, N, seq, ahead, bad);
}

}  // extern "C"
