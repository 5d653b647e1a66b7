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

// Cache rows per block = passes x 256.  A large launch walks CHUNK rows per block (the query's match masks are built once per
// block); an explorer-size one -- a DyNA-PPO environment step: ten queries against the 3000 sequences seen so far -- had 30 blocks
// on 256 CUs that way, each thread walking four pairs in turn: one pass per block gives it four times the blocks (distance matrix
// of that step 139 -> 59 us).
static inline int fx_dist_passes(int64_t C, int64_t Q) { return ((C + CHUNK - 1) / CHUNK) * Q < 512 ? 1 : CHUNK / 256; }

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

// Small launches (fx_dist_passes == 1): the block's 256 cache rows -- ONE contiguous, 16-byte aligned piece of the cache -- are
// copied to LDS with 16-byte loads that are all in flight at once, and the recurrence reads its text bytes from LDS.  Read from
// global memory, a thread's row costs one ~0.45 us round trip PER COLUMN (the byte is needed to pick the match mask; nothing of the
// next column can start before it): 42 us for a 90-residue pair whose arithmetic is ~5 us.  A large launch hides that behind its
// other waves (0.94 of the integer-VALU roofline at 2000 x 20 000, BASELINE's shape) and keeps reading global memory.
constexpr int FX_STAGE_MAX_L = 160;      // 40 KiB of rows beside the match masks
__device__ __forceinline__ void stage_rows(const uint8_t* __restrict__ cache, int64_t c_first, int64_t C, int L, uint8_t* ts) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int64_t rows = C - c_first < 256 ? C - c_first : 256;
    const int total = (int)rows * L, n16 = total >> 4;
    const uint8_t* src = cache + c_first * L;               // (c_first is a multiple of 256: 16-byte aligned whatever L)
    constexpr int MAXV = FX_STAGE_MAX_L / 16;               // 16-byte pieces per thread
    u4 v[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) { const int i = (int)threadIdx.x + k * 256; if (i < n16) v[k] = reinterpret_cast<const u4*>(src)[i]; }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) { const int i = (int)threadIdx.x + k * 256; if (i < n16) reinterpret_cast<u4*>(ts)[i] = v[k]; }
    for (int i = (n16 << 4) + (int)threadIdx.x; i < total; i += 256) ts[i] = src[i];
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

template <int W, typename Word, bool STAGE>
__global__ void __launch_bounds__(256) k_min_dist(int mode, const uint8_t* __restrict__ q, const uint8_t* __restrict__ cache,
                                                  int64_t C, int L, unsigned long long* __restrict__ keys, int passes) {
    __shared__ Word peq[256 * W];
    __shared__ uint8_t qs[W * 8 * sizeof(Word)];
    __shared__ unsigned long long wave_min[4];
    extern __shared__ __attribute__((aligned(16))) uint8_t fx_text_rows[];
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * passes * 256;    // `passes` x 256 cache rows per block (fx_dist_passes)
    if (STAGE) stage_rows(cache, c0, C, L, fx_text_rows);     // (passes == 1; published by load_query's barriers)
    const int m = load_query<W, Word>(q, qi, L, qs, peq);

    unsigned long long best = ~0ull;
    for (int k = 0; k < passes; ++k) {
        const int64_t c = c0 + k * 256 + tid;
        if (c >= C) break;
        const int d = pair_distance<W, Word>(mode, m, L, qs, peq, STAGE ? (const uint8_t*)fx_text_rows + tid * L : cache + c * L);
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
template <int W, typename Word, bool STAGE>
__global__ void __launch_bounds__(256) k_distances(int mode, const uint8_t* __restrict__ q, const uint8_t* __restrict__ cache,
                                                   int64_t C, int L, uint8_t* __restrict__ out, int passes) {
    __shared__ Word peq[256 * W];
    __shared__ uint8_t qs[W * 8 * sizeof(Word)];
    extern __shared__ __attribute__((aligned(16))) uint8_t fx_text_rows[];
    const int64_t qi = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * passes * 256;
    if (STAGE) stage_rows(cache, c0, C, L, fx_text_rows);     // (passes == 1; published by load_query's barriers)
    const int m = load_query<W, Word>(q, qi, L, qs, peq);
    for (int k = 0; k < passes; ++k) {
        const int64_t c = c0 + k * 256 + threadIdx.x;
        if (c >= C) break;
        const int d = pair_distance<W, Word>(mode, m, L, qs, peq, STAGE ? (const uint8_t*)fx_text_rows + threadIdx.x * L : cache + c * L);
        out[qi * C + c] = (uint8_t)(d > 255 ? 255 : d);
    }
}

// ---- distances up to a BOUND (round 4): min(distance, K + 1) -----------------------------------------------------------------
// `sequence_density` (dyna_ppo.py:106-114) only asks which stored sequences are within 2 edits and how far exactly.  Ukkonen's band:
// a distance <= K can only run through cells |i - j| <= K of the DP matrix, 2 K + 1 per text column, and once a whole band column
// exceeds K the answer is "more" -- for unrelated sequences after a handful of columns, where the bit-parallel recurrence above
// always walks all of them (38 us for a 90-residue pair on a lone wave).  One thread per pair, the band in registers.
// (the band itself: myers.h fx_bounded_distance, shared with the host test hook fx_debug_bounded_distance)
template <int K, bool STAGE>
__global__ void __launch_bounds__(256) k_distances_bounded(int mode, const uint8_t* __restrict__ q, const uint8_t* __restrict__ cache,
                                                           int64_t C, int L, uint8_t* __restrict__ out) {
    __shared__ uint8_t qs[FX_MINDIST_MAX_L];
    __shared__ int m_s;
    extern __shared__ __attribute__((aligned(16))) uint8_t fx_text_rows[];
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.y, c0 = (int64_t)blockIdx.x * 256;
    if (STAGE) stage_rows(cache, c0, C, L, fx_text_rows);
    for (int i = tid; i < L; i += 256) qs[i] = q[qi * L + i];
    __syncthreads();
    if (tid == 0) {
        int m = L;
        for (int i = 0; i < L; ++i) if (qs[i] == 0) { m = i; break; }
        m_s = m;
    }
    __syncthreads();
    const int64_t c = c0 + tid;
    if (c >= C) return;
    out[qi * C + c] = (uint8_t)fx_bounded_distance<K>(mode == FX_HAMMING, m_s, L, qs, STAGE ? (const uint8_t*)fx_text_rows + tid * L : cache + c * L);
}

// ---- patterns longer than 768 symbols: the bit-parallel recurrence in STRIPS of 12 words -------------------------
// `editdistance.eval` (noisy_abstract_model.py:51) takes strings of any length.  A strip is 768 rows of the DP matrix
// (what fits a thread's registers); strip s is run over all text columns with the pattern rows [768 s, 768 s + 768):
// its top boundary is the bottom boundary of strip s - 1 -- one horizontal delta in {-1, 0, +1} per column, kept in a
// per-thread column of a global scratch array ([column][thread]: coalesced) -- its left boundary the usual +1 vertical
// deltas, and the distance is m plus the bottom deltas of the LAST strip.  The workgroup shares the query, so the
// strip loop and the rebuild of the match-mask table are workgroup-uniform.  Same per-block arithmetic as myers.h.
constexpr int LW = 12;                                      // words per strip (myers.h fx_myers_strip)

// MATRIX = false: fold (remapped distance, index) into keys[q]; true: dense uint8 distances
template <bool MATRIX>
__global__ void __launch_bounds__(256) k_min_dist_long(int mode, const uint8_t* __restrict__ q, int64_t q0, const uint8_t* __restrict__ cache,
                                                       int64_t C, int L, unsigned long long* __restrict__ keys, uint8_t* __restrict__ out,
                                                       int8_t* __restrict__ hbuf) {
    __shared__ uint64_t peq[256 * LW];
    __shared__ unsigned long long wave_min[4];
    __shared__ int m_s;
    const int tid = threadIdx.x;
    const int64_t qi = q0 + blockIdx.y;
    const uint8_t* qrow = q + qi * L;
    if (tid == 0) {
        int m = L;
        for (int i = 0; i < L; ++i) if (qrow[i] == 0) { m = i; break; }
        m_s = m;
    }
    __syncthreads();
    const int m = m_s;
    const size_t hstride = (size_t)gridDim.x * gridDim.y * 256;
    int8_t* hcol = hbuf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
    const int nstrips = m > 0 ? (m + 64 * LW - 1) / (64 * LW) : 1;
    unsigned long long best = ~0ull;
    const int64_t c0 = (int64_t)blockIdx.x * CHUNK;
    for (int k = 0; k < CHUNK / 256; ++k) {
        const int64_t c = c0 + k * 256 + tid;
        const bool live = c < C;                            // (every thread walks the strips: the table rebuild is a barrier)
        const uint8_t* t = cache + (live ? c : 0) * L;
        int d = 0;
        if (mode == FX_HAMMING) {
            if (live) for (int i = 0; i < L; ++i) d += (t[i] != qrow[i]);
        } else {
            for (int s = 0; s < nstrips; ++s) {
                const int r0 = s * 64 * LW, rows = (m - r0) < 64 * LW ? (m - r0) : 64 * LW;
                __syncthreads();                            // everybody is done with the previous table
                uint64_t mk[LW];
#pragma unroll
                for (int w = 0; w < LW; ++w) mk[w] = 0;
                for (int i = 0; i < rows; ++i)
                    if (qrow[r0 + i] == tid) {
#pragma unroll
                        for (int w = 0; w < LW; ++w)
                            if ((i >> 6) == w) mk[w] |= 1ull << (i & 63);
                    }
#pragma unroll
                for (int w = 0; w < LW; ++w) peq[tid * LW + w] = mk[w];
                __syncthreads();
                if (live) {
                    const int part = fx_myers_strip<LW>(rows > 0 ? rows : 0, L, [&](int ch, int w) { return peq[ch * LW + w]; },
                                                        [&](int i) { return (int)t[i]; }, (const signed char*)hcol, (signed char*)hcol,
                                                        hstride, s == 0, s == nstrips - 1);
                    if (s == nstrips - 1) d = m + part;
                }
            }
        }
        if (!live) continue;
        if (MATRIX) {
            out[(qi - q0) * C + c] = (uint8_t)(d > 255 ? 255 : d);
        } else {
            const unsigned dp = d == 1 ? 0u : (d == 0 ? 1u : (unsigned)d);
            const unsigned long long key = ((unsigned long long)dp << 32) | (unsigned long long)c;
            best = key < best ? key : best;
        }
    }
    if (MATRIX) return;
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

__global__ void k_min_dist_finish(const unsigned long long* __restrict__ keys, int64_t Q, int32_t* __restrict__ dist,
                                  int64_t* __restrict__ arg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const unsigned long long k = keys[i];
    const unsigned dp = (unsigned)(k >> 32);
    dist[i] = dp == 0u ? 1 : (dp == 1u ? 0 : (int32_t)dp);
    arg[i] = (int64_t)(k & 0xffffffffull);
}

// Long rows: queries in batches sized so that the boundary scratch ([column][thread] int8) stays below 256 MiB.
template <bool MATRIX>
int launch_long(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C, int L,
                unsigned long long* d_keys, uint8_t* d_out) {
    const int64_t chunks = (C + CHUNK - 1) / CHUNK;
    const int64_t per_query = chunks * 256 * (int64_t)L;
    int64_t qb = ((int64_t)256 << 20) / per_query;
    qb = qb < 1 ? 1 : (qb > Q ? Q : qb);
    void* hbuf = nullptr;
    if (int rc = fx_scratch(e, 4, (size_t)(per_query * qb), &hbuf)) return rc;
    for (int64_t q0 = 0; q0 < Q; q0 += qb) {
        const int64_t qn = Q - q0 < qb ? Q - q0 : qb;
        hipLaunchKernelGGL(k_min_dist_long<MATRIX>, dim3((unsigned)chunks, (unsigned)qn), dim3(256), 0, e->stream, mode, d_q, q0, d_cache, C, L,
                           d_keys, MATRIX ? d_out + q0 * C : nullptr, (int8_t*)hbuf);
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

}  // namespace

int fx_launch_min_dist(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C,
                       int L, unsigned long long* d_keys) {
    if (Q == 0 || C == 0) return FX_OK;
    if (Q > 65535) return fx_fail(e, FX_EINVAL, "min_dist: more than 65535 queries per call (split the batch)");
    FX_HIP(e, hipMemsetAsync(d_keys, 0xFF, sizeof(unsigned long long) * (size_t)Q, e->stream));
    if (L > FX_MINDIST_MAX_L) return launch_long<false>(e, mode, d_q, Q, d_cache, C, L, d_keys, nullptr);
    const int passes = fx_dist_passes(C, Q);
    const bool stage = passes == 1 && L <= FX_STAGE_MAX_L && e->dist_stage;      // small launch: the block's cache rows through LDS
    const size_t lds = (size_t)256 * L + 16;
    dim3 grid((unsigned)((C + (int64_t)passes * 256 - 1) / ((int64_t)passes * 256)), (unsigned)Q), block(256);
    if (L <= 32) {          // one 32-bit word per column: half the integer work of the 64-bit form
        { if (stage) hipLaunchKernelGGL((k_min_dist<1, uint32_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<1, uint32_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); }
    } else {
        switch ((L + 63) / 64) {
            case 1: { if (stage) hipLaunchKernelGGL((k_min_dist<1, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<1, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            case 2: { if (stage) hipLaunchKernelGGL((k_min_dist<2, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<2, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            case 3: { if (stage) hipLaunchKernelGGL((k_min_dist<3, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<3, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            case 4: { if (stage) hipLaunchKernelGGL((k_min_dist<4, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<4, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            case 5: case 6: { if (stage) hipLaunchKernelGGL((k_min_dist<6, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<6, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            case 7: case 8: { if (stage) hipLaunchKernelGGL((k_min_dist<8, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<8, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;
            default: { if (stage) hipLaunchKernelGGL((k_min_dist<12, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); else hipLaunchKernelGGL((k_min_dist<12, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_keys, passes); } break;   // L <= 768 (full-length AAV capsid: 735)
        }
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

int fx_launch_distances(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C,
                        int L, uint8_t* d_out) {
    if (Q == 0 || C == 0) return FX_OK;
    if (Q > 65535) return fx_fail(e, FX_EINVAL, "distances: more than 65535 queries per call");
    if (L > FX_MINDIST_MAX_L) return launch_long<true>(e, mode, d_q, Q, d_cache, C, L, nullptr, d_out);
    const int passes = fx_dist_passes(C, Q);
    const bool stage = passes == 1 && L <= FX_STAGE_MAX_L && e->dist_stage;      // small launch: the block's cache rows through LDS
    const size_t lds = (size_t)256 * L + 16;
    dim3 grid((unsigned)((C + (int64_t)passes * 256 - 1) / ((int64_t)passes * 256)), (unsigned)Q), block(256);
    if (L <= 32) {
        { if (stage) hipLaunchKernelGGL((k_distances<1, uint32_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<1, uint32_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); }
    } else {
        switch ((L + 63) / 64) {
            case 1: { if (stage) hipLaunchKernelGGL((k_distances<1, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<1, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            case 2: { if (stage) hipLaunchKernelGGL((k_distances<2, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<2, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            case 3: { if (stage) hipLaunchKernelGGL((k_distances<3, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<3, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            case 4: { if (stage) hipLaunchKernelGGL((k_distances<4, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<4, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            case 5: case 6: { if (stage) hipLaunchKernelGGL((k_distances<6, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<6, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            case 7: case 8: { if (stage) hipLaunchKernelGGL((k_distances<8, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<8, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;
            default: { if (stage) hipLaunchKernelGGL((k_distances<12, uint64_t, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out, passes); else hipLaunchKernelGGL((k_distances<12, uint64_t, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out, passes); } break;   // L <= 768 (full-length AAV capsid: 735)
        }
    }
    FX_HIP(e, hipGetLastError());
    return FX_OK;
}

// min(distance, K + 1) for K = 1 .. 3 and rows of at most 768 bytes: FX_EUNSUPPORTED otherwise (the caller takes the exact matrix).
int fx_launch_distances_bounded(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C, int L,
                                int K, uint8_t* d_out) {
    if (Q == 0 || C == 0) return FX_OK;
    if (K < 1 || K > 3 || L > FX_MINDIST_MAX_L || Q > 65535 || !e->dist_bounded) return FX_EUNSUPPORTED;
    const bool stage = L <= FX_STAGE_MAX_L && e->dist_stage;
    const size_t lds = stage ? (size_t)256 * L + 16 : 0;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)Q), block(256);
#define FX_BOUNDED(KK)                                                                                                                   \
    { if (stage) hipLaunchKernelGGL((k_distances_bounded<KK, true>), grid, block, lds, e->stream, mode, d_q, d_cache, C, L, d_out);     \
      else hipLaunchKernelGGL((k_distances_bounded<KK, false>), grid, block, 0, e->stream, mode, d_q, d_cache, C, L, d_out); }
    if (K == 1) FX_BOUNDED(1) else if (K == 2) FX_BOUNDED(2) else FX_BOUNDED(3)
#undef FX_BOUNDED
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
