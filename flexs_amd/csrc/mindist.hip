// K4 min_dist: NoisyAbstractModel._get_min_distance (noisy_abstract_model.py:42-60)
// for Q queries against C cache keys (insertion order), integer-exact.
//
// grid = (cache chunks, queries).  A block builds the query's 256-entry match-mask
// table in LDS once, every thread then runs the bit-parallel Levenshtein
// (myers.h) -- or a byte-compare Hamming -- against its cache entries and the
// block folds (remapped distance, cache index) with a 64-bit min:
//     key = (d' << 32) | index,   d' = 0 if d == 1, 1 if d == 0, d otherwise
// which is exactly the reference's loop: the FIRST entry at distance 1 wins
// outright (early `return`, :53-54), otherwise the first strict minimum (:56-58).
// Rows are L bytes; a sequence shorter than L is NUL-padded (editdistance.eval takes strings of any
// two lengths, noisy_abstract_model.py:51): the pattern length is the query's first NUL, the text stops
// at the cache row's first NUL.  Hamming compares the padded rows byte-wise (a pad against a letter
// counts as a mismatch).
// Integer VALU bound; HBM traffic is the cache once per query row (L2-resident).
#include "fx_common.h"
#include "myers.h"

namespace {

constexpr int CHUNK = 1024;     // cache entries per block
constexpr int FX_MINDIST_MAX_L = 768;   // 12 words of 64 pattern rows

// Block prologue shared by both kernels: stage query `qi` in LDS, let thread c build the match masks of byte
// value c, and return the query length (its first NUL, every thread finds it itself).
template <int W, typename Word>
__device__ __forceinline__ int load_query(const uint8_t* __restrict__ q, int64_t qi, int L, uint8_t* qs, Word* peq) {
    constexpr int BITS = 8 * (int)sizeof(Word);
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += 256) qs[i] = q[qi * L + i];
    __syncthreads();
    int m = L;
    Word mk[W];
#pragma unroll
    for (int w = 0; w < W; ++w) mk[w] = 0;
    for (int i = 0; i < L; ++i) {
        const int ch = qs[i];
        if (ch == 0) { m = i; break; }                 // NUL-padded (ragged) query
        if (ch == tid) {
#pragma unroll
            for (int w = 0; w < W; ++w)
                if ((i / BITS) == w) mk[w] |= Word(1) << (i % BITS);
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) peq[tid * W + w] = mk[w];
    __syncthreads();
    return m;
}

// Distance of the staged query to one cache row.
template <int W, typename Word>
__device__ __forceinline__ int pair_distance(int mode, int m, int L, const uint8_t* qs, const Word* peq,
                                             const uint8_t* __restrict__ t) {
    if (mode == FX_HAMMING) {
        int d = 0;
        for (int i = 0; i < L; ++i) d += (t[i] != qs[i]);
        return d;
    }
    return fx_myers_distance<W, true, Word>(
        m, L, [&](int ch, int w) { return peq[ch * W + w]; }, [&](int i) { return (int)t[i]; });
}

template <int W, typename Word>
__global__ void __launch_bounds__(256) k_min_dist(int mode, const uint8_t* __restrict__ q, const uint8_t* __restrict__ cache,
                                                  int64_t C, int L, unsigned long long* __restrict__ keys) {
    __shared__ Word peq[256 * W];
    __shared__ uint8_t qs[W * 8 * sizeof(Word)];
    __shared__ unsigned long long wave_min[4];
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.y;
    const int m = load_query<W, Word>(q, qi, L, qs, peq);

    unsigned long long best = ~0ull;
    const int64_t c0 = (int64_t)blockIdx.x * CHUNK;
    for (int k = 0; k < CHUNK / 256; ++k) {
        const int64_t c = c0 + k * 256 + tid;
        if (c >= C) break;
        const int d = pair_distance<W, Word>(mode, m, L, qs, peq, cache + c * L);
        const unsigned dp = d == 1 ? 0u : (d == 0 ? 1u : (unsigned)d);
        const unsigned long long key = ((unsigned long long)dp << 32) | (unsigned long long)c;
        best = key < best ? key : best;
    }
    // wave min, then block min, then one atomic
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0) wave_min[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = wave_min[0];
        for (int w = 1; w < 4; ++w) b = wave_min[w] < b ? wave_min[w] : b;
        if (b != ~0ull) atomicMin(&keys[qi], b);
    }
}

// Dense Q x C distance matrix (uint8, clamped) -- same per-pair code as k_min_dist.
template <int W, typename Word>
__global__ void __launch_bounds__(256) k_distances(int mode, const uint8_t* __restrict__ q, const uint8_t* __restrict__ cache,
                                                   int64_t C, int L, uint8_t* __restrict__ out) {
    __shared__ Word peq[256 * W];
    __shared__ uint8_t qs[W * 8 * sizeof(Word)];
    const int64_t qi = blockIdx.y;
    const int m = load_query<W, Word>(q, qi, L, qs, peq);
    const int64_t c0 = (int64_t)blockIdx.x * CHUNK;
    for (int k = 0; k < CHUNK / 256; ++k) {
        const int64_t c = c0 + k * 256 + threadIdx.x;
        if (c >= C) break;
        const int d = pair_distance<W, Word>(mode, m, L, qs, peq, cache + c * L);
        out[qi * C + c] = (uint8_t)(d > 255 ? 255 : d);
    }
}

__global__ void k_min_dist_finish(const unsigned long long* __restrict__ keys, int64_t Q, int32_t* __restrict__ dist,
                                  int64_t* __restrict__ arg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const unsigned long long k = keys[i];
    const unsigned dp = (unsigned)(k >> 32);
    dist[i] = dp == 0u ? 1 : (dp == 1u ? 0 : (int32_t)dp);
    arg[i] = (int64_t)(k & 0xffffffffull);
}

}  // namespace

int fx_launch_min_dist(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C,
                       int L, unsigned long long* d_keys) {
    if (Q == 0 || C == 0) return FX_OK;
    if (L > FX_MINDIST_MAX_L) return fx_fail(e, FX_EUNSUPPORTED, "min_dist: sequence length > 768");
    if (Q > 65535) return fx_fail(e, FX_EINVAL, "min_dist: more than 65535 queries per call (split the batch)");
    FX_HIP(e, hipMemsetAsync(d_keys, 0xFF, sizeof(unsigned long long) * (size_t)Q, e->stream));
    dim3 grid((unsigned)((C + CHUNK - 1) / CHUNK), (unsigned)Q), block(256);
    if (L <= 32) {          // one 32-bit word per column: half the integer work of the 64-bit form
        hipLaunchKernelGGL((k_min_dist<1, uint32_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys);
    } else {
        switch ((L + 63) / 64) {
            case 1: hipLaunchKernelGGL((k_min_dist<1, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            case 2: hipLaunchKernelGGL((k_min_dist<2, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            case 3: hipLaunchKernelGGL((k_min_dist<3, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            case 4: hipLaunchKernelGGL((k_min_dist<4, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            case 5: case 6: hipLaunchKernelGGL((k_min_dist<6, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            case 7: case 8: hipLaunchKernelGGL((k_min_dist<8, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;
            default: hipLaunchKernelGGL((k_min_dist<12, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys); break;   // L <= 768 (full-length AAV capsid: 735)
        }
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_distances(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C,
                        int L, uint8_t* d_out) {
    if (Q == 0 || C == 0) return FX_OK;
    if (L > FX_MINDIST_MAX_L) return fx_fail(e, FX_EUNSUPPORTED, "distances: sequence length > 768");
    if (Q > 65535) return fx_fail(e, FX_EINVAL, "distances: more than 65535 queries per call");
    dim3 grid((unsigned)((C + CHUNK - 1) / CHUNK), (unsigned)Q), block(256);
    if (L <= 32) {
        hipLaunchKernelGGL((k_distances<1, uint32_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out);
    } else {
        switch ((L + 63) / 64) {
            case 1: hipLaunchKernelGGL((k_distances<1, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            case 2: hipLaunchKernelGGL((k_distances<2, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            case 3: hipLaunchKernelGGL((k_distances<3, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            case 4: hipLaunchKernelGGL((k_distances<4, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            case 5: case 6: hipLaunchKernelGGL((k_distances<6, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            case 7: case 8: hipLaunchKernelGGL((k_distances<8, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;
            default: hipLaunchKernelGGL((k_distances<12, uint64_t>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); break;   // L <= 768 (full-length AAV capsid: 735)
        }
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_min_dist_finish(fx_engine* e, const unsigned long long* d_keys, int64_t Q, int64_t C, int32_t* d_dist,
                              int64_t* d_arg) {
    (void)C;
    if (Q == 0) return FX_OK;
    hipLaunchKernelGGL(k_min_dist_finish, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, e->stream, d_keys, Q, d_dist, d_arg);
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}
