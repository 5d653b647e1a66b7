// Device helpers shared by the MFMA scoring kernels (gfx950 only).
//
// Formulation.  Every layer is computed TRANSPOSED:  Out^T[ch_out][seq] =
// W^T[ch_out][ch_in] * In^T[ch_in][seq], with v_mfma_f32_16x16x4_f32:
//   A operand (16 x 4)  = weights^T   lane l holds A[i = l & 15][k = l >> 4]
//   B operand (4 x 16)  = activations lane l holds B[k = l >> 4][j = l & 15]   (j = sequence)
//   C/D       (16 x 16)               lane l, reg r holds D[row = 4*(l >> 4) + r][col = l & 15]
// so a lane always owns ONE sequence (l & 15) and the lane group g = l >> 4
// owns channels 16*tile + 4*g + r.  Feeding accumulator register r of tile mi
// straight back as the B operand of k-step (mi, r) of the next layer contracts
// over channels {16*mi + 4*g + r : g = 0..3} -- the weights are pre-permuted on
// the host to match (pack.cpp), so activations never leave registers between
// layers: no LDS round trip, no transposes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ReLU on the float bit pattern: a negative float is a negative integer, a non-negative one orders like its bits,
// so max_i32(bits, 0) is relu(x) in exactly one VALU instruction (fmaxf makes the compiler quieten some MFMA
// results first: an extra v_max_f32 v, v, v per component).  -0.0 -> +0.0; a NaN stays NaN or becomes 0 depending
// on its sign bit and ends as 0 through nan_to_num either way.
__device__ __forceinline__ float relu1(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ f4 relu4(f4 v) {
    f4 r;
    r.x = relu1(v.x); r.y = relu1(v.y); r.z = relu1(v.z); r.w = relu1(v.w);
    return r;
}
// Running global max of relu(x): `pool` is >= +0 throughout, so comparing the float bit patterns as signed
// integers gives max(pool, x) -- any negative x (sign bit set) loses, non-negative floats order like integers --
// in ONE v_max_i32 per component.  fmaxf would cost a second VALU instruction per component to quieten the MFMA
// result first (IEEE mode), and VALU instructions cost matrix-pipe time here.  A NaN
// activation wins the comparison and ends as 0 through nan_to_num, which is what np.max + np.nan_to_num give.
__device__ __forceinline__ f4 pool_max4(f4 pool, f4 x) {
    f4 r;
    r.x = __int_as_float(max(__float_as_int(pool.x), __float_as_int(x.x)));
    r.y = __int_as_float(max(__float_as_int(pool.y), __float_as_int(x.y)));
    r.z = __int_as_float(max(__float_as_int(pool.z), __float_as_int(x.z)));
    r.w = __int_as_float(max(__float_as_int(pool.w), __float_as_int(x.w)));
    return r;
}

// XCD-aware work split.  Workgroup b is observed to run on XCD b % 8, and every XCD has its own L2: with the plain
// "block b takes the b-th slice of the (member, tile) units" the 32 workgroups behind one L2 would be spread over
// every ensemble member, and each of the 8 L2s would fetch every member's weights from HBM.  This bijection hands
// XCD x a CONTIGUOUS range of slices instead, so the workgroups that share an L2 also share (mostly) one member's
// weight image and neighbouring rows of the sequence batch.  Placement is a speed assumption only.
__device__ __forceinline__ unsigned fx_xcd_block() {
    const unsigned b = blockIdx.x, g = gridDim.x, per = g >> 3, rem = g & 7u, x = b & 7u;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}

// Work split of the persistent scoring kernels.  Units = (member, tile), member-major; every workgroup takes one
// contiguous range.  Whenever the grid has at least M workgroups the range lies inside ONE member: member m owns
// the workgroups [ceil(G m / M), ceil(G (m + 1) / M)) (contiguous, so with fx_xcd_block they share an L2) and its
// tiles are cut evenly over them.  A range that straddled two members made that workgroup wait at the member
// boundary for its slowest wave, refill LDS and start over -- one tile duration plus a fill, which with few tiles
// per workgroup WAS the kernel's tail (profiles/archive/r2_trace_probe: 3 members x 1e4 sequences, 38 us span of which 12
// were two straddling workgroups; 194 -> ~180 us on the 3 x 1e5 bench launch).
// spread = true (relay launches, FxRelay): plain block order instead -- member 0's workgroups, the only ones that pull rows over
// PCIe, then sit on all eight XCDs instead of one.
// Round 6, measured and NOT kept: cutting a workgroup's share in whole rounds of four tiles (72 or 76 instead of 73 / 74, so that a 19th
// tile always finds four waves on its SIMD).  The in-kernel timeline (profiles/r6_trace_probe.json) shows workgroups of 72 tiles leaving at
// 175 us and those with a 19-tile SIMD at 192 us, but the 76-tile workgroups leave at 192 us as well: +0.4 % on the headline launch, within
// +-1 % on seven other shapes (profiles/r6_unit_quant_ab.log).
__device__ __forceinline__ void fx_unit_range(int64_t TG, int M, int64_t& u_lo, int64_t& u_hi, bool spread = false) {
    const int64_t G = gridDim.x, bid = spread ? (int64_t)blockIdx.x : (int64_t)fx_xcd_block();
    const int64_t U = (int64_t)M * TG;
    // ... unless cutting at the member boundaries makes the largest share bigger: with coarse units (the two-waves-per-
    // tile protein kernel: a tile is a ~1.4 ms serial walk, 12 tiles per workgroup) one more tile on the workgroups of a
    // member with fewer workgroups outweighs a boundary crossing (3 x 1024 tiles on 256 workgroups: 12 each, or 13 on 85)
    const int64_t nb_min = M > 0 ? G / M : 0;
    const bool by_member = M > 1 && nb_min >= 1 && (TG + nb_min - 1) / nb_min <= (U + G - 1) / G;
    if (by_member) {
        const int64_t m = bid * M / G;
        const int64_t g_lo = (m * G + M - 1) / M, g_hi = ((m + 1) * G + M - 1) / M;
        const int64_t nb = g_hi - g_lo, j = bid - g_lo;
        u_lo = m * TG + TG * j / nb;
        u_hi = m * TG + TG * (j + 1) / nb;
    } else {
        u_lo = U * bid / G;
        u_hi = U * (bid + 1) / G;
    }
}

// The SIMD this wave runs on (HW_REG_HW_ID bits 5:4).  The matrix pipe belongs to the SIMD, so a workgroup's tiles are
// dealt to its four SIMDs in equal shares and only the waves OF a SIMD pull from that share (one LDS counter per SIMD):
// with one shared counter the share of a SIMD was whatever its waves happened to grab -- on mid-size launches (1-2
// tiles per wave) one SIMD of a CU ran 8 tiles while another ran 4, and the kernel ended with the slowest SIMD
// (profiles/archive/r2_trace_probe: MLP, 1e5 sequences: median wave done at 35 us, last at 46 us).
__device__ __forceinline__ int fx_simd_id() { return (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); }

// Distinct issue priorities for the waves that share a SIMD (by hardware wave slot), for the whole kernel: the waves of
// a workgroup start their first tiles in the same cycle and would otherwise sit in the same phase together -- all in
// the LDS-bound first layer with the matrix pipe idle, then all queueing for the pipe.  With static priorities the
// first wave runs ahead and the phases of the four waves interleave from the first tile on (engine option "wave_prio").
__device__ __forceinline__ void fx_stagger_priority() {
    switch (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3) {       // HW_REG_HW_ID.WAVE_ID (slot in the SIMD)
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
}

// The 16 x L sequence bytes of a tile are contiguous in memory: one wave copies them into its own LDS scratch with 16-byte
// loads (ONE vector-memory instruction for L <= 64 instead of one byte-load instruction per position and lane group),
// and the lanes then pick their bytes with ds_read_u8.  At kernel start every wave of the machine asks for its first
// tile at once; with byte loads that was tens of thousands of requests and a ~2 us round trip per dependent step of a
// first layer (profiles/archive/r2_trace_probe: 8 us before the first MFMA of an MLP tile).  `src` may be unaligned (a row
// offset into the caller's buffer): gfx950 runs compute with unaligned access enabled; the scratch is 16-byte aligned.
typedef const __attribute__((address_space(3))) uint8_t* fx_lds_u8p;
struct __attribute__((packed, aligned(1))) FxBytes16 { uint32_t w[4]; };
__device__ __forceinline__ void fx_stage_tile(const uint8_t* __restrict__ src, int bytes, uint8_t* dst_lds, int lane) {
    for (int off = lane * 16; off < bytes; off += 64 * 16) {
        if (off + 16 <= bytes) {
            const FxBytes16 v = *reinterpret_cast<const FxBytes16*>(src + off);
            uint32_t* d = reinterpret_cast<uint32_t*>(dst_lds + off);
            d[0] = v.w[0]; d[1] = v.w[1]; d[2] = v.w[2]; d[3] = v.w[3];
        } else {
            for (int b = off; b < bytes; ++b) dst_lds[b] = src[b];
        }
    }
}

// Relay forms of fx_stage_tile (FxRelay): PASS = the tile's bytes also go to `relay` in device memory with device-scope stores (the
// caller then waits for them -- fx_wait_vm(0) -- and raises the tile's flag); FROM = they come from there with device-scope loads.
// A tile starts at a multiple of 128 bytes in both areas: the 8-byte accesses are aligned.
__device__ __forceinline__ void fx_stage_tile_pass(const uint8_t* __restrict__ src, int bytes, uint8_t* dst_lds, int lane, uint8_t* relay) {
    for (int off = lane * 16; off < bytes; off += 64 * 16) {
        if (off + 16 <= bytes) {
            const FxBytes16 v = *reinterpret_cast<const FxBytes16*>(src + off);
            uint32_t* d = reinterpret_cast<uint32_t*>(dst_lds + off);
            d[0] = v.w[0]; d[1] = v.w[1]; d[2] = v.w[2]; d[3] = v.w[3];
            unsigned long long* r8 = reinterpret_cast<unsigned long long*>(relay + off);
            __hip_atomic_store(r8, (unsigned long long)v.w[0] | ((unsigned long long)v.w[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(r8 + 1, (unsigned long long)v.w[2] | ((unsigned long long)v.w[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // (the ragged last tile of a batch: byte by byte -- no register array indexed at run time)
            for (int b = off; b < bytes; ++b) {
                const uint8_t x = src[b];
                dst_lds[b] = x;
                __hip_atomic_store(relay + b, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
__device__ __forceinline__ void fx_stage_tile_from(const uint8_t* relay, int bytes, uint8_t* dst_lds, int lane) {
    for (int off = lane * 16; off < bytes; off += 64 * 16) {
        const unsigned long long* r8 = reinterpret_cast<const unsigned long long*>(relay + off);
        const unsigned long long a = __hip_atomic_load(r8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(r8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t* d = reinterpret_cast<uint32_t*>(dst_lds + off);
        d[0] = (uint32_t)a; d[1] = (uint32_t)(a >> 32); d[2] = (uint32_t)b; d[3] = (uint32_t)(b >> 32);
    }
}

// Share of the tile range [t_lo, t_hi) that belongs to this wave's SIMD, proportional to the number of the workgroup's
// waves each SIMD hosts (counted once at kernel start, fx_count_simd_wave: 4-4-4-4 for a 16-wave workgroup, but
// nothing here depends on the placement -- a SIMD without waves simply gets no tiles).
struct FxSimdShare { int before, mine, total; };
// counters: 4 ints of LDS that were zeroed before the workgroup's last barrier; the caller puts one barrier between
// fx_count_simd_wave (every wave) and fx_simd_share (two existing barriers of the kernels are used, none is added)
__device__ __forceinline__ void fx_count_simd_wave(int* counters, int simd) {
    if ((threadIdx.x & 63) == 0) atomicAdd(&counters[simd], 1);
}
__device__ __forceinline__ FxSimdShare fx_simd_share(const int* counters, int simd) {
    FxSimdShare r{0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int w = counters[i];
        if (i < simd) r.before += w;
        if (i == simd) r.mine = w;
        r.total += w;
    }
    // wave-uniform: keep the three counts (and the tile bounds derived from them) in scalar registers
    r.before = __builtin_amdgcn_readfirstlane(r.before);
    r.mine = __builtin_amdgcn_readfirstlane(r.mine);
    r.total = __builtin_amdgcn_readfirstlane(r.total);
    return r;
}

// Workgroup copy of a member's packed weights into LDS.  Eight 16-byte loads are in flight per thread before the
// first LDS store, so the ~100 KiB image costs a couple of L2 round trips instead of one per 16 bytes per thread
// (which is what a plain copy loop compiles to, and what small calls and small batches then mostly wait for).
template <int DEPTH = 8>
__device__ __forceinline__ void fill_lds(f4* __restrict__ dst, const f4* __restrict__ src, int n4) {
    const int bd = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n4; i0 += DEPTH * bd) {
        f4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (i0 + k * bd < n4) v[k] = src[i0 + k * bd];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (i0 + k * bd < n4) dst[i0 + k * bd] = v[k];
    }
}

// ---- direct global -> LDS copies (gfx950: global_load_lds_dword / _dwordx4) ---------------------------------------
// The data never passes through VGPRs, so a wave can put its whole share of a member's weight image in flight at
// kernel start and begin computing as soon as the FIRST part (what the first layers read) has landed, while the rest
// is still on its way: a small launch otherwise waits ~3-5 us for ~100-150 KiB before its first instruction of work
// (profiles/archive/r2_trace_probe).  Written as inline assembly on purpose: the compiler's own tracking of such loads puts an
// `s_waitcnt vmcnt(0)` in front of the next LDS read that might alias, i.e. waits for everything.  Ordering is by the
// wave's vector-memory counter: loads return in order, so `fx_wait_vm(n)` (at most n still in flight) after issuing
// part 1 and then n loads of part 2 guarantees part 1 -- loads the compiler issues in between only make the wait
// stricter, never weaker.  Each wave waits for ITS loads; a workgroup barrier then publishes the part to all waves.
// LDS address of lane i = M0 + i * (4 | 16): M0 is saved and restored around the instruction.
__device__ __forceinline__ unsigned fx_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void fx_dma16(const void* g_lane, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g_lane), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void fx_dma4(const void* g_lane, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g_lane), "s"(lds_base) : "memory");
}
// A FULL tile's bytes (a multiple of 16) global -> this wave's LDS scratch without registers and without waiting: the caller scores
// another tile meanwhile and runs fx_wait_vm(0) before it reads the scratch.  For rows in HOST memory (zero-copy launches): the
// PCIe round trip of a tile's bytes then lies beside the previous tile's work instead of in front of its own.
__device__ __forceinline__ void fx_stage_tile_dma(const uint8_t* src, int bytes, uint8_t* dst_lds, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane(fx_lds_addr(dst_lds));
    for (int c = 0; c * 1024 < bytes; ++c) {
        const int off = c * 1024 + lane * 16;
        if (off < bytes) fx_dma16(src + off, base + (unsigned)c * 1024u);
    }
}
// Relay (FxRelay): a tile that sits in this wave's LDS scratch goes on to the relay area (device-scope stores; the caller waits --
// fx_wait_vm(0) -- and raises the tile's flag).
__device__ __forceinline__ void fx_relay_from_lds(const uint8_t* src_lds, int bytes, int lane, uint8_t* relay) {
    for (int off = lane * 16; off < bytes; off += 64 * 16) {
        const uint32_t* d = reinterpret_cast<const uint32_t*>(src_lds + off);
        const unsigned long long a = (unsigned long long)d[0] | ((unsigned long long)d[1] << 32), b = (unsigned long long)d[2] | ((unsigned long long)d[3] << 32);
        unsigned long long* r8 = reinterpret_cast<unsigned long long*>(relay + off);
        __hip_atomic_store(r8, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r8 + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// at most n (wave-uniform) vector-memory loads of this wave still in flight
__device__ __forceinline__ void fx_wait_vm(int n) {
    switch (n) {
#define FX_VM_CASE(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        FX_VM_CASE(1) FX_VM_CASE(2) FX_VM_CASE(3) FX_VM_CASE(4) FX_VM_CASE(5) FX_VM_CASE(6) FX_VM_CASE(7) FX_VM_CASE(8)
        FX_VM_CASE(9) FX_VM_CASE(10) FX_VM_CASE(11) FX_VM_CASE(12) FX_VM_CASE(13) FX_VM_CASE(14) FX_VM_CASE(15) FX_VM_CASE(16)
        FX_VM_CASE(17) FX_VM_CASE(18) FX_VM_CASE(19) FX_VM_CASE(20) FX_VM_CASE(21) FX_VM_CASE(22) FX_VM_CASE(23) FX_VM_CASE(24)
#undef FX_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// ---- 16-byte stores / loads that are coherent across the XCDs WITHOUT fences (agent scope: `sc1`) ------------------------
// Each XCD has its own L2; data one workgroup hands to workgroups on other XCDs inside a launch normally needs an agent-scope
// release (a write-back of the XCD's whole L2) and an acquire (an invalidate) around the hand-off -- ~0.5 us each, and the
// fences of one XCD serialise (profiles/r4_mailbox_probe3.log): 243 workgroups x 2 fences were 21 us of a 64 us launch.
// An sc1 store writes through to memory and an sc1 load misses the non-coherent levels, so the hand-off needs only the
// counter that orders it.  Inline assembly: the compiler has no 128-bit atomics, and it does not count these operations --
// the caller waits with fx_wait_vm(0) before it signals (stores) / the loads wait themselves.
__device__ __forceinline__ void fx_store16_agent(f4* p, f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
// eight loads in flight, one wait
__device__ __forceinline__ void fx_load16x8_agent(const f4* p0, const f4* p1, const f4* p2, const f4* p3, const f4* p4, const f4* p5,
                                                  const f4* p6, const f4* p7, f4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
                 : "memory");
}

// Workgroup copy of n4 16-byte words (1 KiB chunks dealt round-robin to the `waves` waves); returns the number of
// loads THIS wave issued (wave-uniform).  dst must be 16-byte aligned LDS, src 16-byte aligned global memory.
__device__ __forceinline__ int fx_dma_fill(float* dst_lds, const float* __restrict__ src, int n4, int waves) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned base = __builtin_amdgcn_readfirstlane(fx_lds_addr(dst_lds));
    int issued = 0;
    for (int c = wave * 64; c < n4; c += waves * 64) {
        if (c + lane < n4) fx_dma16(reinterpret_cast<const f4*>(src) + c + lane, base + (unsigned)c * 16u);
        ++issued;
    }
    return issued;
}

// The 256-byte character LUT, global -> LDS without a round trip of its own: as a plain copy it is a load the workgroup
// waits for BEFORE it asks for its weights, one more dependent ~1 us step at the start of every launch.  The caller
// runs fx_wait_vm(0) (or waits for any younger load) before the barrier that publishes the weights.
__device__ __forceinline__ void fx_lut_dma(uint8_t* lut_s, const uint8_t* __restrict__ lut) {
    if (threadIdx.x < 64) fx_dma4(lut + threadIdx.x * 4, __builtin_amdgcn_readfirstlane(fx_lds_addr(lut_s)));
}

// In-kernel timeline (engine option "trace", debugging / profiling only): the first lane of every wave stamps the
// constant-rate wall clock (100 MHz) into slot `slot` of its row; `t` is null in normal operation.
#define FX_TRACE_SLOTS 16
#define FX_TRACE_WAVES 16
#if defined(FX_TRACE)
__device__ __forceinline__ void fx_stamp(unsigned long long* t, int slot, unsigned long long v = ~0ull) {
    if (t && (threadIdx.x & 63) == 0)
        t[((size_t)blockIdx.x * FX_TRACE_WAVES + (threadIdx.x >> 6)) * FX_TRACE_SLOTS + slot] = (v == ~0ull) ? wall_clock64() : v;
}
// end of a tile: first-tile end (3), last-tile end (4), tiles processed (5)
#define FX_TILE_DONE() do { if (p.trace) { if (tiles_done == 0) fx_stamp(p.trace, 3); fx_stamp(p.trace, 4); fx_stamp(p.trace, 5, ++tiles_done); } } while (0)
#else
// production build: the timeline costs registers in kernels that have none to spare (the 16-wave forms sit at their
// 128-VGPR budget), so the stamps exist only in `make trace` builds (-DFX_TRACE, libflexs_amd_trace.so)
__device__ __forceinline__ void fx_stamp(unsigned long long*, int, unsigned long long = 0) {}
#define FX_TILE_DONE() do { } while (0)
#endif

// Phase stamps INSIDE a tile (slots 8..10 of the first tile) split the tile's code into scheduling regions -- the
// layers no longer overlap, +7 % on the GlobalEpistasis kernel -- so they exist only in builds made with
// -DFX_TRACE_PHASES as well (make trace-phases).
#if defined(FX_TRACE) && defined(FX_TRACE_PHASES)
#define FX_PHASE_STAMP(slot) do { if (tiles_done == 0) fx_stamp(p.trace, (slot)); } while (0)
#else
#define FX_PHASE_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ f4 splat4(float v) { f4 r = {v, v, v, v}; return r; }

__device__ __forceinline__ float fx_nan_to_num(float v) {
    // np.nan_to_num (keras_model.py:77)
    if (v != v) return 0.f;
    if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return v;
}

// acc[mo][nt] += W-blocks(mi, mo) applied to in[mi][nt], for all mi < TI, mo < TO.
// wblk points at block (mi = 0, mo = 0); blocks are ordered mi-major, 64 f4 each.
// Loop order (mi, r, mo, nt) keeps >= TO*NT independent MFMAs between two
// accumulations into the same register quad (16x16x4 f32: 32-cycle issue,
// 40-cycle dependent latency).
// rl_last (wave-uniform, 1..4): number of k-steps of the LAST input tile that carry real
// channels (hidden-unit tail laid out k-step-major by the packer, fx_hidden_pos); the
// remaining k-steps would multiply zeros and are skipped.  (Hidden sizes that were rounded up
// to a larger instantiated tile count pass 4: their padding tiles are computed as zeros.)
// PRIO: raise the wave's issue priority around each MFMA cluster.  Measured +2 % on the unrolled L = 8
// kernel in an interleaved A/B (profiles/archive/r1_run16_setprio_ab.md) but neutral-to-negative on the dynamic-loop,
// MLP and GE kernels (profiles/archive/r1_run17_*), so it is opt-in per instantiation.
template <int TI, int TO, int NT, bool PRIO = false, typename WPtr>
__device__ __forceinline__ void mma_layer(WPtr wblk, const f4 (&in)[TI][NT], f4 (&acc)[TO][NT], int lane,
                                          int rl_last = 4) {
#pragma unroll
    for (int mi = 0; mi < TI; ++mi) {
        f4 a[TO];
#pragma unroll
        for (int mo = 0; mo < TO; ++mo) a[mo] = wblk[(mi * TO + mo) * 64 + lane];
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mi == TI - 1 && r >= rl_last) break;
#pragma unroll
            for (int mo = 0; mo < TO; ++mo)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mo][nt] = mfma16(a[mo][r], in[mi][nt][r], acc[mo][nt]);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
}

// Weight matrices that do not fit LDS (hidden sizes 129..256: TI*TO KiB per layer): instead of every wave streaming all blocks from L2
// for every tile (which makes the kernel L2-bandwidth-bound), the WORKGROUP streams them once per round of WAVES tiles through a
// double-buffered LDS slab of KG input tiles (KG*TO KiB): one barrier per slab, all waves of the workgroup in lockstep.
// Rounds 1-5 carried the next slab through registers (16-byte pieces in flight during a slab's MFMAs, parked in the other buffer
// afterwards).  Round 6: the slabs are copied global -> LDS DIRECTLY (fx_dma16: no registers, no ds_write behind a slab's MFMAs) and the
// stream of slabs runs on ACROSS layers and rounds: while a layer's last slab is being multiplied, the first slab of the NEXT layer
// (`wnext`: the round's other H x H layer, or the next round's first one; nullptr = nothing follows) is already on its way into the free
// buffer, so only the first layer of a member's first round waits for a trip to L2.  `st` carries the buffer the layer starts in and
// whether its first slab has been asked for.  Same MFMA sequence per accumulator as before (and as mma_layer): the same bits;
// MLP H=200 1e5 / 1e6: 171.2 -> 167.6 / 1568 -> 1530 us (profiles/r6_slab_dma_ab.log).
struct FxSlabStream { int par; bool primed; };
template <int TI, int TO, int KG, int WAVES>
__device__ __forceinline__ void fx_slab_issue(const f4* __restrict__ wglob, int sl, f4* buf, int wave, int lane) {
    constexpr int SLAB = KG * TO * 64;
    const int tiles = TI - sl * KG < KG ? TI - sl * KG : KG;   // input tiles of this slab
    const int chunks = tiles * TO;                              // 1 KiB (one wave-wide 16-byte copy) each
    const unsigned base = __builtin_amdgcn_readfirstlane(fx_lds_addr(buf));
    const f4* src = wglob + (size_t)sl * SLAB + lane;
    for (int c = wave; c < chunks; c += WAVES) fx_dma16(src + c * 64, base + (unsigned)c * 1024u);
}
template <int TI, int TO, int KG, int WAVES>
__device__ __forceinline__ void mma_layer_slab_dma(const f4* __restrict__ wglob, const f4* __restrict__ wnext, f4* slab, const f4 (&in)[TI][1],
                                                   f4 (&acc)[TO][1], int lane, int rl_last, FxSlabStream& st) {
    constexpr int SLAB = KG * TO * 64;                         // f4 per (full) slab
    constexpr int NS = (TI + KG - 1) / KG;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (!st.primed) {
        __syncthreads();                                        // every wave is done with both buffers
        fx_slab_issue<TI, TO, KG, WAVES>(wglob, 0, slab + st.par * SLAB, wave, lane);
    }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of slab sl have landed ...
        __syncthreads();                                        // ... everybody's have; and the other buffer is free (slab sl - 1 is done)
        f4* nbuf = slab + (st.par ^ 1) * SLAB;
        if (sl + 1 < NS) fx_slab_issue<TI, TO, KG, WAVES>(wglob, sl + 1, nbuf, wave, lane);
        else if (wnext) fx_slab_issue<TI, TO, KG, WAVES>(wnext, 0, nbuf, wave, lane);
        const f4* buf = slab + st.par * SLAB;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const int mi = sl * KG + j;
            if (mi < TI) {
                constexpr int MC = 4;                           // (as in mma_layer_slab)
#pragma unroll
                for (int m0 = 0; m0 < TO; m0 += MC) {
                    f4 a[MC];
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        if (m0 + c < TO) a[c] = buf[(j * TO + m0 + c) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mi == TI - 1 && r >= rl_last) break;
#pragma unroll
                        for (int c = 0; c < MC; ++c)
                            if (m0 + c < TO) acc[m0 + c][0] = mfma16(a[c][r], in[mi][0][r], acc[m0 + c][0]);
                    }
                }
            }
        }
        st.par ^= 1;
    }
    st.primed = wnext != nullptr;
}

// acc[mo][nt] = bias[16*mo + 4*g .. +3] broadcast over the lane's sequence
template <int TO, int NT, typename BPtr>
__device__ __forceinline__ void init_bias(BPtr bias, f4 (&acc)[TO][NT], int g) {
#pragma unroll
    for (int mo = 0; mo < TO; ++mo) {
        const f4 b = *reinterpret_cast<const f4*>(&bias[16 * mo + 4 * g]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mo][nt] = b;
    }
}

template <int T, int NT>
__device__ __forceinline__ void relu_tiles(f4 (&x)[T][NT]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) x[t][nt] = relu4(x[t][nt]);
}

// final H -> 1 layer: out[seq] = b + sum_ch w[ch] * h[ch][seq].  Each lane sums its
// 4*HT channels, the four lane groups are combined with two cross-lane adds.
template <int HT, int NT, typename VPtr>
__device__ __forceinline__ void final_dot(VPtr wv, float bout, const f4 (&h)[HT][NT], float (&out)[NT], int g) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) out[nt] = 0.f;
#pragma unroll
    for (int mo = 0; mo < HT; ++mo) {
        const f4 w = *reinterpret_cast<const f4*>(&wv[16 * mo + 4 * g]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            out[nt] = fmaf(w.x, h[mo][nt].x, out[nt]);
            out[nt] = fmaf(w.y, h[mo][nt].y, out[nt]);
            out[nt] = fmaf(w.z, h[mo][nt].z, out[nt]);
            out[nt] = fmaf(w.w, h[mo][nt].w, out[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        out[nt] += __shfl_xor(out[nt], 16);
        out[nt] += __shfl_xor(out[nt], 32);
        out[nt] += bout;
    }
}
