// Internal declarations shared by the translation units of libflexs_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/flexs_amd.h"

#define FX_MAX_M 16          // models fused per launch (larger ensembles are split)
#define FX_ERR_BADCHAR 1u    // bit in the deferred device error word
#define FX_LP_BAR_BYTES (19 * 128)   // barrier counters of the layer-parallel protein form: top + 16 groups, a 128-byte line each; line 17: units finished (completion flag); line 18: a pre-launched instance's go / leave decision
#define FX_ERR_TIMEOUT 2u    // a device-side barrier (layer-parallel protein form) was not passed in time
#define FX_ERR_STARVED 4u    // a launched-first host call (FxRowsReady) never saw some of its rows arrive: the host redoes the call the plain way

// "Launch first, pack behind" (round 5): a big host call of list[str] launches its kernels BEFORE the strings are packed; the packing
// threads fill the pinned staging area stage by stage -- stage j = the 16-row tiles t with t % Q == j -- and publish, per packing
// lane, how many stages they have finished (word = base + stages done, device memory written through the BAR).  A wave that pulls
// tile t waits until every lane has published stage t % Q.  The staging area of such a call is TILE-PITCHED: tile t's rows start at
// t * pitch, a whole number of 128-byte lines, so no cache line holds rows of two tiles and a line fetched for one tile can never
// carry a neighbour's rows from before they were packed -- which is what lets the wait do without a cache invalidate (an acquire
// fence of agent or system scope after the poll cost the 1024 waves of a launch ~85 us, and reading the rows with system-scope
// loads instead cost the MLP launch 50-120 us: profiles/r5_launch_first.log).
// words == nullptr: the call's rows are all there, N x L contiguous (every other call).
// Relay inside a launched-first launch of an ENSEMBLE whose members would each read the rows over PCIe again (M x the bytes: what made
// zero-copy ruinous for 8 x GlobalEpistasis L = 90): member 0's workgroups read a tile from the host staging area, score it, and
// pass its bytes on to device memory (`dst`, same tile pitch; device-scope stores, then flags[tile] = seq); the other members' waves
// wait for the tile's flag and read it from there (device-scope loads).  The rows cross PCIe ONCE and the upload runs beside the
// scoring instead of in front of it.  flags == nullptr: no relay.
struct FxRelay {
    uint8_t* dst;
    unsigned* flags;            // one word per tile
    unsigned seq;               // this call's value (never 0)
    int pitch;                  // bytes from one tile to the next in dst: whole 128-byte lines (no line holds bytes of two tiles)
    int spread;                 // 1 = workgroups take their unit ranges in plain block order: member 0's sit on all eight XCDs
};

struct FxRowsReady {
    const unsigned* words;      // one unsigned per lane, all in one 64-byte line (one request per poll)
    unsigned base;              // this call's origin of the counts (stale values of earlier calls compare as "nothing yet")
    int lanes;                  // 1 .. 16
    int Q;                      // number of stages
    int pitch;                  // bytes from one tile's rows to the next tile's
};
// The resident form (score_cnn_quad.hip / score_dense_small.hip, SERVER).  Round 3: <= 16 tile slots per member on a third
// of the CUs, one tile per slot and request -> 256 sequences.  Round 4 (engine option serve_wide): a generation may take
// most of the chip -- up to FX_SERVE_TILES slots per member -- and a slot walks the tiles slot, slot + T, slot + 2T, ... of a
// request, so requests of up to FX_SERVE_CAP sequences / FX_SERVE_BYTES bytes (a Random-explorer round of 2001 8-mers,
// CbAS batches, Adalead's root + first-children call) are answered without a launch, a weight fill or a second launch.
#define FX_SERVE_TILES 256     // tile slots per member at most (one resident workgroup each)
#define FX_SERVE_BYTES 65536   // N x seq_len bytes per request at most
#define FX_SERVE_CAP 4096      // sequences per request at most
#define FX_SERVE_LEAVE 0xFFFFull   // request word that tells the resident workgroups to leave (sequence number 0, N = 0xFFFF: neither occurs in a request)
#define FX_SERVE_TINY 0x4000ull            // request word, bit 14: the request's bytes (<= FX_SERVE_TINY_BYTES) ride in the request word's own 64-byte line (FxMailIn::tiny)
#define FX_SERVE_TINY_BYTES 48
#define FX_SERVE_STREAM 0x8000ull          // request word, bit 15 of the row count: the bytes follow the request (FxMailIn::ready)
#define FX_SERVE_FAST 16       // the first slots of every member poll without a pause (explorer-size calls); the others sleep between polls

// ---------------------------------------------------------------- shapes
struct FxShape {
    int kind, L, A, F, H, K;
    int K3() const { return A - 1; }
    int L1() const { return L - K + 1; }
};

// Packed (MFMA fragment) layout of one model's weights; all offsets in floats.
// A "block" is 64 lanes x 4 floats: block[lane][r] = W[kin(lane, r)][16*mo + (lane & 15)],
// i.e. the A operand (weights^T, M = output channels) of four consecutive
// v_mfma_f32_16x16x4_f32 k-steps, fetched with one ds_read_b128 per lane.
//   dense-style blocks (conv2/conv3 taps, dense layers; input = a previous MFMA result tile mi):
//       kin = 16*mi + 4*(lane >> 4) + r        (k-step r consumes accumulator register r)
//   first-layer blocks (one-hot input; conv1 or MLP layer 1; step group sg):
//       kin = 16*sg + 4*r + (lane >> 4)        (k-step s = 4*sg + r covers rows 4s .. 4s+3)
#define FX_PAIR_PAD 4          // floats between the MLP's pair rows (LDS bank spread, pack.cpp)

// Rows of the conv1 gather table are 16 FT + 4 floats apart (round 6): sixteen lanes gather sixteen rows at once, and with rows 32 floats = 128 bytes
// apart every row starts in one of TWO 16-byte slots of the 256-byte bank row (4- to 8-way conflicts on every gather of the protein CNN: 35 % of its
// LDS cycles, profiles/r6_pmc_targets.md); 36 floats apart the rows of a tap walk all sixteen slots.  MLP rows (16 HT floats) are a different table.
#define FX_C1_ROW(FT) (16 * (FT) + 4)
struct FxPackLayout {
    int FT, HT;                 // output tiles of 16: filters, hidden units (HT rounded up to an instantiated size)
    int HTR;                    // hidden tiles that hold real units = ceil(H / 16) <= HT; the rest is zero padding
    int SG1;                    // first-layer step groups = ceil(rows / 16)
    int RLH;                    // k-steps that carry real channels in the LAST hidden tile (1..4), see fx_hidden_pos
    int64_t off_first;          // SG1 x (FT|HT) blocks
    int64_t off_c2, off_c3;     // K x FT x FT, K3 x FT x FT blocks           (CNN)
    int64_t off_cb;             // b1[16FT] b2[16FT] b3[16FT]                 (CNN)
    int64_t off_w1p;            // conv1 kernel, plain [K*A] rows of 16 FT floats, FX_C1_ROW(FT) floats apart (gather form of the one-hot conv)
    int64_t conv_floats;        // everything above (the part that must sit in LDS)
    int64_t off_d1, off_d2, off_d3;  // dense blocks: CNN d1 FTxHT, d2 HTxHT; MLP d2, d3 HTxHT; GE d3 HTxHT
    int64_t off_db;             // bias / vector area (layout per kind, see pack.cpp)
    int64_t total_floats;       // end of the vector area = end of what the kernels may stage in LDS as one image
    // MLP on a 4-letter alphabet: first-layer rows summed for PAIRS of positions, [(L/2)*16 + (L%2)*4][16HT] floats after
    // the image (row (pi*16 + 4*c0 + c1) = row(2pi, c0) + row(2pi+1, c1)); -1 = none
    int64_t off_w1pair, pair_floats;   // MLP, 4 letters: pre-summed rows per pair of positions, FX_PAIR_PAD floats of padding per row
    int64_t alloc_floats;       // size of the packed buffer (total_floats + pair table)
};

FxPackLayout fx_pack_layout(const FxShape& s);
// Position of hidden unit h inside the zero-padded 16*HT layout.  Hidden units are
// interchangeable, so the tail tile (H - 16*(HT-1) real units) is laid out k-step-major
// (unit i -> lane group i % 4, register i / 4): its real channels then occupy only
// ceil(tail / 4) of the 4 k-steps, and the layers that CONSUME the tile skip the rest
// (H = 100: 25 instead of 28 k-steps per HxH layer).
int fx_hidden_pos(int h, int H);
int64_t fx_num_params(const FxShape& s);
// MFMA instructions issued per 16-sequence tile per member by the MFMA kernels (pack.cpp); -1 = no MFMA kernel
int64_t fx_mfma_per_tile(const FxShape& s);
// Keras get_weights() blob -> packed fragment layout (host only, no device needed).
void fx_pack_weights(const FxShape& s, const float* blob, float* packed);

// Mailboxes of the resident ("server") form (host and device share these declarations).  Neither direction reads over
// PCIe: the request lives in DEVICE memory, the host stores into it through the BAR (write-combining on the host side:
// sfence after the bytes and after the request word) and the resident workgroups poll it locally; the answers live in
// pinned HOST memory, the workgroups store (score, tag) pairs + a system fence and the host spins on its own memory.
// Measured (tools/probes/mailbox_probe2.hip): 2.7 us round trip for one workgroup, 6 us for 48, against 25 us for 48
// workgroups polling a request word in host memory (their reads serialise on the PCIe link).
struct FxMailIn {                                      // device memory (fine-grained); the host only ever WRITES it
    alignas(64) unsigned long long req;                // (sequence number << 16) | number of sequences; written LAST by the host
    // TINY requests (round 4; request word bit 14): a request of at most 48 sequence bytes -- one to six 8-mers, three 14-mers: 85 %
    // of Adalead's calls -- carries them in the rest of the request word's line, with the word once more at the line's end.  The
    // first slot of every member polls the whole line (sixteen lanes, one 64-byte read) and has the bytes in registers the
    // moment it sees the request: the second dependent read of device memory (the byte area, ~0.8 us) is gone from the call.
    unsigned char tiny[FX_SERVE_TINY_BYTES];
    unsigned long long req_tail;                       // = req of the tiny request (a reader that finds req != req_tail reads again)
    alignas(64) unsigned stop;                         // host: 1 = leave now
    alignas(64) unsigned char bytes[FX_SERVE_BYTES];   // the request's sequences, row-major
    // the same request word once more, for the tile slots beyond the first few of every member (wide generation): a word that
    // hundreds of workgroups poll is a queue in front of ONE memory channel, and the few slots an explorer-size request needs
    // would wait in it (12.5 vs 11.2 us per 20-sequence call with 240 pollers on `req`, profiles/r4_server_wide_ab_first.log).
    // The many poll this copy -- far from `req`'s line, with a pause between polls; it is written just BEFORE `req`.
    alignas(64) unsigned long long req_wide;
    // STREAMED requests (round 4; request word bit 15, FX_SERVE_STREAM): the request is posted BEFORE its bytes, and the host
    // packs the caller's strings straight into `bytes`, raising `ready` = (sequence number << 16) | rows packed so far every few
    // hundred rows -- a tile starts as soon as ITS rows are there, so the packing of a 2001-string call (4.6 us) runs beside the
    // first tiles instead of in front of them.  A workgroup waits for its rows in fx_server_rows_ready.
    alignas(64) unsigned long long ready;
};
static_assert(offsetof(FxMailIn, stop) == 64 && offsetof(FxMailIn, bytes) == 128, "server_start clears the first 128 bytes");
static_assert(offsetof(FxMailIn, tiny) == 8 && offsetof(FxMailIn, req_tail) == 56, "the request line: word, 48 bytes, word");
// Mailbox of a PRE-LAUNCHED instance of the layer-parallel protein form (round 4; score_cnn_pair.hip, fx_resident.hip "armed"): fine-grained
// device memory the host stores into through the BAR.  The instance has its weights in LDS and waits for ITS request word --
// sixteen copies, a line each: block b polls copy b & 15 -- then reads the request's sequences from `bytes`.
// request word of instance s for N sequences: (s << 16) | N; (s << 16) | 0xFFFF tells instance s to leave (words of other instances are ignored)
struct FxLpMail {
    struct alignas(64) Word { unsigned long long w; } req[16];
    alignas(64) unsigned char bytes[FX_SERVE_BYTES];
};
struct FxMailOut {                                     // pinned host memory; the host only ever READS it (after zeroing it between generations)
    alignas(64) volatile unsigned alive[FX_MAX_M][FX_SERVE_TILES];   // 1 while the workgroup of (member, tile slot) is resident
    // one 8-byte store per (member, sequence): the score's bits and the request's sequence number (bit 31: the tile met a
    // character outside the alphabet).  Every element validates itself: no flag, no ordering assumption between lines
    alignas(64) volatile unsigned long long ans[FX_MAX_M][FX_SERVE_CAP];
};


// ---------------------------------------------------------------- handles
struct fx_engine {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // chunked host call (fx_score_begin / _submit / _finish): transfers ride a second stream, so that the upload of piece
    // k + 1 and the download of piece k - 1 overlap the kernels of piece k; one event triple per piece
    hipStream_t copy_stream = nullptr;
    static constexpr int MAX_PIECES = 32;
    hipEvent_t ev_in[MAX_PIECES] = {}, ev_done[MAX_PIECES] = {}, ev_out[MAX_PIECES] = {};
    std::string last_error;
    // engine-side counters (fx_engine_counters): what went through this engine since creation / the last reset
    struct Counters {
        int64_t host_calls = 0;        // fx_score / fx_score_finish / fx_decode_score calls (host buffers in, host scores out)
        int64_t device_calls = 0;      // fx_score_dev / fx_score_planes_dev launches (device buffers)
        int64_t sequences = 0;         // sequences scored (x members = forwards)
        int64_t forwards = 0;          // sequences x members
        int64_t bytes_h2d = 0, bytes_d2h = 0;   // copy-path bytes of the host calls
        int64_t zero_copy_calls = 0;   // host calls whose kernels read / wrote pinned host memory directly
        int64_t pair_evals = 0;        // (query, cache entry) distance evaluations
        int64_t train_steps = 0;       // mini-batch steps x members run by fx_train_fit
    } counters;
    // deferred error word (device) + pinned host mirror
    unsigned* d_err = nullptr;
    unsigned* h_err = nullptr;
    std::map<void*, size_t> result_bufs;   // fx_result_alloc: registered anonymous mappings (address -> length), freed by fx_result_free only
    // cached device LUT
    uint8_t* d_lut = nullptr;
    uint8_t h_lut[256];
    bool lut_valid = false;
    // growable scratch
    void* d_scratch[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (5: the relay area of a plain host call, FxRelay)
    size_t scratch_bytes[6] = {0, 0, 0, 0, 0, 0};
    void* d_train = nullptr;      // fx_train_fit arena (grown on demand, kept between fits)
    size_t train_bytes = 0;
    int64_t train_prof_ns[5] = {};   // the last fx_train_fit: ns since entry at "image filled", "upload enqueued", "launches enqueued", "synchronised", "results copied out"
    void* h_train = nullptr;      // pinned host image of the arena's uploaded regions (train.hip)
    size_t train_host_bytes = 0;
    void* d_zero_pool = nullptr;  // fx_zero_pool: all-zero between launches (the kernels that use it clean up after themselves)
    size_t zero_pool_bytes = 0;
    void* h_pinned[2] = {nullptr, nullptr};
    size_t pinned_bytes[2] = {0, 0};
    // options
    int64_t force_generic = 0;
    int64_t cnn_variant = 0;    // 0 = auto
    int64_t grid_blocks = 0;    // 0 = auto (one per CU)
    int64_t cnn_conv1_mfma = 0; // 1 = one-hot conv1 on MFMA instead of the LDS gather (A/B knob)
    int64_t poison_outputs = 0; // test knob: fill score buffers with NaN before launching, so an element no kernel wrote is caught
    int64_t cnn_pair = 1;       // 1 = wide alphabets (A = 20) use the two-waves-per-tile kernel (score_cnn_pair.hip)
    int64_t cnn_big_units = 12; // work units per CU from which the A = 4 CNN path switches to 16-wave (unrolled) workgroups
    int64_t cnn_seg = -1;       // A = 4 CNN kernel, small batches: -1 = waves of a workgroup split one tile's positions when L1 >= 24, 0 = never, 1 = whenever the batch is small
    int64_t dense_small = 1;    // 1 = explorer-size MLP launches (one tile per workgroup, at most one workgroup per CU) deal a tile's output tiles to 8 waves (score_dense_small.hip); 2 = at any size (test knob); 0 = off
    int64_t cnn_seg_multi = 1;  // 1 = the position-segmented 4-letter form may spread a tile over several (4- or 8-wave) workgroups
    int64_t cnn_pair_seg4 = 1;  // 1 = the segmented protein form may use 4-wave workgroups (one wave per SIMD) when twice as many still fit in one wave of the grid
    int64_t cnn_pair_seg = -1;  // pair kernel, small batches: -1 = segment a tile's positions automatically, 0 = never, n > 0 = force SB = n workgroups per tile
    int64_t dense_slab_coop = 3; // slab form: up to this many leftover tiles (tiles mod 8) of a workgroup are walked by its 8 waves together instead of a whole lockstep round; 0 = off
    int64_t dense_slab = 1;     // MLP / GE with H > 128: HxH blocks staged through LDS slabs by the workgroup (0 = every wave streams them from L2)
    int64_t ge_bytetab = 1;     // 1 = GlobalEpistasis layer 1 gathers from the byte-indexed per-position table (0 = LUT + code-indexed table: A/B)
    int64_t dense_waves = 0;    // MLP / GE canonical hidden width: 0 = auto (8 waves per workgroup for mid-size launches, else 16), 8 / 16 = force
    int64_t dense_few_waves_below = 0;    // auto: tiles per SIMD below which the 8-wave form is used (0 = never: measured no gain, profiles/archive/r2_dense_waves_ab.log)
    int64_t cnn_head_slab = 1;  // 4-letter CNN with more than 128 hidden units, batch launches (>= 16 tiles per workgroup): conv-only kernel + a head kernel that streams the H x H layer through LDS slabs
                                // once per lockstep round of 8 tiles (k_cnn_head_slab) instead of the fused kernel, whose waves stream it from L2 per tile; 0 = the fused kernel (rounds 1-5)
    int64_t cnn_quad_tail = 1;  // K1, unrolled seq_len = 8 form: the (tiles mod 4) last tiles of a workgroup are walked by wave quads in one round behind the main loop; 0 = one wave per tile throughout (rounds 1-5)
    int64_t wave_prio = 1;      // 1 = static, distinct issue priorities for the waves of a SIMD (fx_stagger_priority); 0 = A/B baseline
    int64_t trace = 0;          // 1 = the MFMA scoring kernels stamp an in-kernel timeline into d_trace (fx_debug_trace_read)
    unsigned long long* d_trace = nullptr;
    int64_t cnn_lp = 1;         // 1 = small batches of the canonical protein CNN (A = 20, kernel size 5, 97-112 hidden units) run LAYER-PARALLEL over the chip (conv2 outputs through device memory, one grid barrier) instead of position segments with recomputed halos (score_cnn_pair.hip k_score_cnn_lp); 0 = the SEG form: A/B
    int64_t cnn_lp_debug = 0;   // profiling aid: the layer-parallel kernel leaves after stage k (1 fill, 2 phase 1, 3 conv3 fill, 4 barrier, 5 phase 2, 6 pool); results are garbage
    unsigned* d_lp_bar = nullptr;   // ... its barrier counter (only ever grows; lp_bar_total = what it reads after the launches enqueued so far)
    unsigned lp_bar_total[17] = {};   // [0] top, [1 + g] group g
    bool lp_launched = false;
    // Completion flag (round 4): a kernel form that can tell when its LAST result has been written stores done_seq into a word
    // of pinned host memory, and a small host call polls that word instead of hipStreamSynchronize -- launch-to-"the host
    // knows" beyond the kernel's own time 11.1 -> 6.0 us (tools/probes/sync_latency_probe.hip).  done_armed: the last launch of
    // the current dispatch does so (cleared before every launch group, set by the launcher).
    unsigned* h_done = nullptr; unsigned* d_done = nullptr;
    unsigned done_seq = 0;
    unsigned done_value = 0;         // last value handed to hipStreamWriteValue32 (h_done[8]): launches without a kernel-side flag
    bool done_armed = false;
    // Pre-launched instance of the layer-parallel form (fx_resident.hip lp_arm / lp_serve_armed): after an explorer-size call of a protein
    // CNN ensemble was answered by k_score_cnn_lp, the NEXT instance is enqueued at once -- it fills its weights and waits for
    // its request word in lp_mail -- so a caller that comes back with the same shape within serve_idle_us pays neither the launch
    // latency nor the weight fill.
    FxLpMail* lp_mail = nullptr;     // (host pointer = device pointer: large BAR)
    bool lp_mail_refused = false;
    unsigned* h_lp_state = nullptr; unsigned* d_lp_state = nullptr;   // pinned host word: (instance sequence << 1) | 1 = "left without a request"
    struct LpArmed {
        bool on = false;
        std::vector<fx_model*> models; std::vector<uint64_t> versions;
        int64_t N = 0; int L = 0; int mode = 0;          // mode: 1 = member planes (mean on the host), 2 = (N, M) matrix
        int64_t stride = 0;
        uint8_t lut[256];
        unsigned seq = 0;                                // the instance's completion-flag value; its request word = (seq << 16) | N
        const void* out_dev = nullptr; const void* scratch2 = nullptr; const void* zero_pool = nullptr;   // buffers baked into the armed launch (round-4 advisor: compared, not assumed)
        std::chrono::steady_clock::time_point t{};
    } lp_armed;
    bool lp_arm_next = false;        // launch_lp: make this launch a pre-launched instance
    int lp_launches = 0;             // launch_lp calls of the current dispatch
    int dispatch_groups = 0;         // launch groups of the current dispatch
    int64_t lp_prelaunch = 1;        // option: 1 = pre-launch the next instance after an explorer-size call of the layer-parallel form (0 = never: A/B)
    int64_t lp_armed_served = 0;     // (read) calls answered by a pre-launched instance
    int64_t done_flag = 1;           // option: 1 = poll the completion flag where a kernel offers one, 0 = always hipStreamSynchronize (A/B)
    int64_t cnn_quad = 1;       // 1 = small launches of the canonical 4-letter CNN with L <= 16 share a tile among four waves (score_cnn_quad.hip); 2 = whatever the size (test knob); 0 = off
    int64_t dma_fill = 1;       // 1 = weight images go global -> LDS directly (global_load_lds), all in flight at kernel start, the first layers start when THEIR part has landed (0 = through registers, whole image before the first tile: A/B)
    int64_t stage_fill = 1;     // 1 = CNN launches with fewer tiles than waves per workgroup load the conv part first and let the idle waves bring the head's weights (0 = whole image before the first tile: A/B)
    int64_t stage_bytes = 1;    // 1 = MLP (pair rows) / GE (byte table) tiles copy their 16 x L sequence bytes into per-wave LDS scratch with 16-byte loads (0 = byte loads from global memory: A/B)
    int64_t mlp_l1_pos = 1;     // MLP whose first-layer rows do not fit LDS (protein alphabets), batch launches: 1 = the first layer position-major by its own kernel (k_mlp_l1_pos, score_dense_l1.h: the rows
                                // cross L2 -> LDS once per 16-32 tiles) + the dense kernel from a scratch; 0 = every sequence gathers its seq_len rows from L2 (rounds 1-5)
    int64_t mlp_l1_pos_tiles = 3; // ... from more than this many tiles per CU on (below: the small-launch form, score_dense_small.hip)
    int64_t mlp_pair = 1;       // 1 = MLP layer 1 on a 4-letter alphabet gathers one pre-summed row per PAIR of positions (0 = one row per position: A/B; 2 = pairs for H <= 128 only, the slab form of wider layers keeps the plain rows: A/B)
    int64_t mlp_l1_mfma = 0;    // 1 = one-hot MLP layer 1 on MFMA instead of the LDS gather (A/B knob)
    int64_t dense_pipe = 0;     // MLP (pair rows) / GE (byte table): 1 / 2 = the software-pipelined form (tile t + 1's first layer inside tile t's MFMA layers, 8 waves, two-part direct LDS fill; 2 = A operands double-buffered by hand).  Bit-identical but measured 11-13 % SLOWER than the 16-wave form at every size (profiles/r3_dense_pipe_ab.log): off; kept as the A/B
    int64_t train_rows = 0;     // fx_train_fit: mini-batch rows per workgroup (0 = auto: 16, or 8 for small batches)
    int64_t train_lds = 2;      // fx_train_fit: 2 = a slice's activations / gradients AND the member's weights live in LDS when they fit, 1 = the workspace only, 0 = global arena (A/B)
    int64_t train_persistent = 0;   // fx_train_fit: 1 = the whole fit is ONE launch when all (slices x members) workgroups are co-resident: two member barriers in device memory per step, Adam by the same workgroups.  Bit-identical to the launch-per-step form and NOT faster: 5.90 vs 5.41 ms for 3 x CNN on 1000 sequences, equal elsewhere (profiles/r4_train_one_launch_ab.log) -- two agent-scope release/acquire barriers + the weight re-read cost what two launch boundaries cost.  Off; kept as the A/B
    int64_t train_split = 0;    // A/B build only: 1 = products with few tiles and a long contraction are cut along the contraction over up to four waves (partial tiles through 16 KiB of LDS, one more barrier).  Bit-identical between instantiations and SLOWER: 24.6 -> 33.1 us per forward+backward launch (profiles/r4_train_split_ab.log)
    int64_t train_canon = 1;    // fx_train_fit: 1 = canonical shapes (CNN(32,100,k5) on 4 letters, MLP(100) on 4 letters, GlobalEpistasis(100) on 20) run the instantiation with compile-time dimensions; 0 = the shape-agnostic code for everything (A/B; bit-identical)
    int64_t train_swizzle = 3;  // fx_train_fit: (3 = any CNN with 32 filters whose weights miss LDS and whose slice has <= 16 M tiles: train_core.h MODE 3 -- paired tiles over rotated kernel rows, register-prefetched staging, sliding-window weight gradient; else as 2) (2 = + dzA over a[2] and conv kernels staged through LDS, train_core.h MODE 2) 1 = CNN fits whose padded workspace misses the LDS budget (long protein sequences) store their position-major arrays with rotated rows (train_core.h) instead of unpadded, 16-way conflicted ones; same bits (CPU); prepared at the end of round 4, not yet measured
    int64_t train_trace = 0;    // profiling aid: 1 = fx_train_fit stamps the phases of the LAST step of member 0, workgroup 0 (fx_debug_train_trace)
    unsigned long long* d_train_dbg = nullptr;
    int64_t train_threads = 0;  // fx_train_fit: threads per forward+backward workgroup (256 / 512 / 1024; 0 = 1024)
    int num_cus = 256;
    int max_lds = 160 * 1024;
    // layout of the score matrix the next launch writes: 0 = row-major (N, M) as the ABI hands it out (np.stack axis=1);
    // > 0 = member-major planes `planar_stride` floats apart (the engine's own intermediate when only the mean is wanted:
    // a unit's 16 scores are then one contiguous 64-byte store instead of 16 four-byte stores 4*M bytes apart)
    int64_t planar_stride = 0;
    // fused ensemble mean: set by the host / device entry points around score_dispatch when only the mean is wanted; a launcher
    // that averages in-kernel (small launches of the canonical CNN: score_cnn_quad.hip) sets fused_mean_done
    float* fuse_mean_out = nullptr;     // (explorer-size form: option fuse_mean)
    float* fuse_mean_batch_out = nullptr;   // (batch form: option fuse_mean_batch)
    bool fused_mean_done = false;
    // resident small-call form (score_cnn_quad.hip / score_dense_small.hip, SERVER): one workgroup per (member, tile slot) stays
    // on the device between explorer-size calls and answers them through the mailboxes above
    struct Server {
        FxMailIn* in = nullptr;                          // device memory; also the host's (write-only) view through the BAR
        FxMailOut* h_out = nullptr;                      // pinned host memory, and the device's view of it
        FxMailOut* d_out = nullptr;
        std::vector<hipStream_t> streams;               // one per group of like members (each group is its own resident launch)
        int groups = 0;
        bool running = false, fresh = false;
        bool wide = false;                               // the running generation's geometry
        std::chrono::steady_clock::time_point t_mid;    // the last request of more than 256 sequences (adaptive geometry)
        int mid_recent = 0;
        std::chrono::steady_clock::time_point t_start, t_post;   // generation start, last request
        unsigned long long seq = 0;
        std::vector<fx_model*> models;
        std::vector<uint64_t> versions;
        uint8_t lut[256] = {};
        int L = 0, cap = 0, wgs = 0;                     // ... and the CUs the generation occupies (one workgroup each)
        int tiles = 0, fast = 0;                         // tile slots per member of the running generation, and how many of them poll `req` (the others: `req_wide`)
        std::vector<fx_model*> refused;                  // the last ensemble that has a member without a resident form
        int refused_L = 0;
        std::vector<fx_model*> pending;                  // the last eligible call's ensemble, and when it came
        std::chrono::steady_clock::time_point t_pending;
        int64_t served = 0, started = 0, fallbacks = 0, fb_info = 0;
        std::chrono::steady_clock::time_point t_entry{};  // entry of the call being served (prof_ns counts from here)
        int64_t posted_N = 0;                            // the posted request's sequences
        int64_t streamed = 0;                            // streamed calls answered
        bool streaming = false; int stream_M = 0;        // a streamed call is open (fx_score_stream_begin .. _end)
        int64_t prof_ns[8] = {};                         // the last served call: checks, request posted, first answer seen, all collected, outputs written (ns since entry)
        int quads[FX_MAX_M] = {};                        // tiles a member's workgroup answers side by side (CNN, wide generation, seq_len <= 8: 3; else 1)
        std::vector<uint8_t> seen;                       // [member][slot]: this generation's workgroup has been seen alive
        std::vector<float> planes;                       // the answers of a request, member-major (host scratch)
    } server;
    bool large_bar = false;     // the host can store into device memory (the resident form needs it)
    int64_t quad_rotate = 1;    // quad CNN form: the waves' roles rotate from quad to quad (balances the MFMA load of a CU's SIMDs; 0 = same roles: A/B)
    int64_t dense_coop = 1;     // MLP / GE persistent kernel: a workgroup's tiles that do not divide among its four SIMDs are walked by groups of 8 waves (0 = one wave each: A/B)
    int64_t serve_idle_us = 500;   // calls of the same ensemble closer than this start / keep the resident workgroups; they leave after twice this long without a request (a device-wide synchronize waits that long for them at most)
    int64_t serve_wide = 1;     // 1 = a resident generation takes (num_cus - serve_reserve_cus) / M tile slots per member and serves requests of up to 4096 sequences, a slot walking several tiles (0 = round 3's geometry: a third of the CUs, <= 16 slots, <= 256 sequences: A/B)
    int64_t serve_reserve_cus = 16;   // CUs a wide generation leaves without a resident workgroup (kernels of other streams -- RCCL, PyTorch -- find room there at once; small ones also fit beside a resident workgroup)
    int64_t call_prof_ns[4] = {};    // the last launched small mean-only host call: ns since entry at "prepared", "launched", "synchronised", "mean taken"
    int64_t host_mean_below = 256;   // launched mean-only host calls of at most this many sequences (zero-copy): member planes to pinned host memory, np.mean's order on the host, no mean launch (0 = the mean kernel: A/B)
    int64_t dist_bounded = 1;   // fx_cache_density: 1 = distances up to the radius by the banded kernel (min(d, radius + 1), radius 1 .. 3), 0 = the exact distance matrix (A/B)
    int64_t dist_stage = 1;     // edit-distance kernels, small launches: 1 = a block's 256 cache rows are copied to LDS and the recurrence reads them there (rows of <= 160 bytes), 0 = every thread reads its row from global memory (A/B)
    int64_t serve_tiny = 1;     // 1 = requests of <= 48 sequence bytes carry them in the request word's own line (0 = always the byte area: A/B)
    int64_t serve_quads = 1;    // wide generation, CNN with seq_len <= 8: tiles per resident workgroup side by side (1 = one; 3 = like the launched form: A/B build only -- slower once requests are streamed, csrc/OPTIONS.md)
    int64_t serve_fence = 0;    // 1 = a system fence after every tile's answers (round 3: ~0.5 us each, and the fences of one XCD serialise -- 24 us for the 378 tiles of a 2001-sequence request, profiles/r4_mailbox_probe3.log); 0 = none: the answers are system-scope stores, which write through by themselves (profiles/r4_mailbox_probe4.log)
    int64_t serve_poll_sleep = 8;     // s_sleep units (64 clocks each) between polls of the slots beyond FX_SERVE_FAST (0 = every slot spins)
    int64_t serve_small = 1;    // 1 = explorer-size calls of canonical CNN ensembles are answered by resident workgroups (0 = a launch per call: A/B)
    int64_t fuse_mean_batch = 0; // A/B build: 1 = batch launches of the 4-letter CNN (rows resident in device memory, mean-only calls) let the LAST member to finish a tile average it in the
                                // scoring kernel (written-through scores, one relaxed device-scope ticket per tile, no fences) instead of launching the mean kernel.  Bit-identical, SLOWER: 183.5 -> 185.1 us per headline step (profiles/r6_fused_mean_ab.log)
    int64_t fuse_mean = 0;      // 1 = explorer-size CNN ensemble calls average in the scoring kernel (last member to finish a tile, tickets + device-scope fences) instead of launching the mean kernel.  Bit-identical, but the two fences cost what the 3 us launch saves: 32.1 vs 32.2 us per call (profiles/r3_fused_mean_ab.log): off, kept as the A/B
    // launched-first host call (fx_score_begin_staged): what the launchers that can wait for rows copy into their arguments, and
    // whether the launch just enqueued did (a launcher that cannot returns FX_EUNSUPPORTED before it enqueues anything)
    struct { bool on = false, used = false; FxRowsReady r = {nullptr, 0, 0, 0, 0}; FxRelay relay = {nullptr, nullptr, 0, 0, 0}; bool relay_used = false; } rows_req;
    int64_t dense_prefetch = 1; // MLP / GE launches that read rows from host memory ask for the next tile's bytes a tile ahead: 1 = in a relay of >= 4 members (where it pays), 2 = always (A/B), 0 = never
    int64_t relay_spread = 0;   // 1 = relay launches take their unit ranges in plain block order: member 0's workgroups on all eight XCDs instead of one (no gain: A/B)
    unsigned* relay_flags = nullptr; size_t relay_flag_words = 0; unsigned relay_seq = 0;
    int64_t launch_relay = 1;   // 1 = launched-first calls of dense ensembles whose plan says "copy" relay the rows through member 0's workgroups (0 = such calls pack, upload, then launch: A/B)
    int64_t launch_relay_calls = 0;
    int64_t rows_min_share = 0;                          // (request) tiles in the shortest per-SIMD share of the launch
    unsigned* rows_words = nullptr;                      // 16 lines of device memory the host stores into (large BAR), or null
    bool rows_refused = false;
    unsigned rows_base = 0;
    bool ascii_host = false;    // (request) the launch about to be enqueued reads its sequences from pinned HOST memory (zero-copy host calls)
    int64_t cnn_stage_host = 1; // 1 = such launches of the canonical 4-letter CNN copy a tile's bytes into LDS with one wide load (0 = a byte load per position: A/B; 2 = 1 + the next tile's bytes asked for a tile ahead: A/B, no gain)
    int64_t launch_first_calls = 0, launch_first_redone = 0;   // (read) host calls launched before their strings were packed; of those, redone the plain way
    int64_t launch_first = 1;   // 1 = big list[str] calls whose kernels can wait for rows are launched before the strings are packed (0 = pack, then launch: A/B)
    // chunked host call in flight (fx_score_begin / _submit / _finish)
    struct {
        bool active = false;
        std::vector<fx_model*> models;
        int64_t N = 0;
        int L = 0;
        bool want_nm = false, want_mean = false;
        uint8_t* h_in = nullptr; uint8_t* d_in = nullptr;
        float* d_nm = nullptr; float* d_mean = nullptr;
        char* h_out = nullptr;
        int pieces = 0;                                  // submitted so far
        bool zero_copy = false;                          // the pieces' kernels read the pinned staging area directly
        int64_t row0[32] = {}, rows[32] = {};
        uint8_t lut[256] = {};      // staged: for a plain second attempt
        unsigned* words = nullptr; unsigned base = 0; int lanes = 0, Q = 0, pitch = 0;
        bool relay = false;         // staged: member 0's workgroups pass the rows on through d_in (FxRelay)
        bool in_place = false;      // staged: the results area is the caller's (fx_result_alloc): finish copies nothing
        bool packed_ok = true;      // staged: the caller packed every row (fx_score_finish_staged says otherwise)
        bool redo = false;          // staged: the launch did not wait for rows after all (never expected), or raised an error word: redo the plain way
        bool staged = false;        // launched first: the kernels are already enqueued and wait for the rows (fx_score_begin_staged)
        int64_t stride = 0;         // staged, mean only: the member planes' stride in d_nm
        unsigned flag[32] = {};     // zero-copy pieces: the value the command processor writes to h_done[8] behind piece k (0 = an event was recorded instead)
    } chunked;
    int64_t chunk_overlap = 0;  // 1 = the chunked host call puts transfers and kernels on two streams (measured SLOWER than one stream: the cross-stream event waits cost more than the overlap buys, profiles/r3_e2e_ab.log; kept as the A/B)
    int64_t zero_copy_bytes = 256 << 10;   // host calls whose input + output are at most this many bytes run zero-copy: the kernels read the sequences from / write the scores to mapped pinned host memory
    int64_t zero_copy_mode = -1;           // host calls beyond zero_copy_bytes: -1 = zero-copy when the plan (fx_plan_host_call) says the PCIe reads hide behind the kernels, 0 = always copy, 1 = always zero-copy (A/B)
};

// The launcher's part of a launched-first call's plan: as many stages as the shortest per-SIMD share has tiles.  false = too few
// to order (nothing launched).
inline bool fx_rows_plan(fx_engine* e) {
    int64_t Q = e->rows_min_share;
    if (Q > 2048) Q = 2048;
    if (Q < 2) return false;
    e->rows_req.r.Q = (int)Q;
    return true;
}

struct fx_model {
    fx_engine* eng = nullptr;
    FxShape shape{};
    FxPackLayout layout{};
    int64_t mfma_per_tile = -1; // fx_mfma_per_tile(shape), once (the host-call planner asks per call: a loop over the positions)
    std::vector<float> blob;    // host copy, Keras order
    float* d_blob = nullptr;    // device copy, Keras order (generic kernels)
    float* d_packed = nullptr;  // device copy, fragment layout (MFMA kernels)
    bool has_weights = false;
    uint64_t version = 0;       // bumped by every fx_model_set_weights
    // GlobalEpistasis first layer as a per-position table indexed by the RAW byte (score_dense_mfma.hip): Lpad x 32
    // floats, tab[l][b - base] = w1[l * A + lut[b]] (0 for bytes outside the alphabet and for the padding rows l >= L);
    // built on the device for the LUT of the call, rebuilt when the weights or the LUT change
    float* d_bytetab = nullptr;
    uint8_t bt_lut[256];
    bool bt_valid = false;
};

struct fx_cache {
    fx_engine* eng = nullptr;
    int L = 0;
    int64_t size = 0, capacity = 0;
    uint8_t* d_keys = nullptr;  // capacity x L bytes, row-major, insertion order
};

struct fx_table {
    fx_engine* eng = nullptr;
    int64_t len = 0;
    double* d_table = nullptr;
};

// ---------------------------------------------------------------- helpers
int fx_fail(fx_engine* e, int status, const std::string& msg);
#define FX_HIP(e, call)                                                                   \
    do {                                                                                  \
        hipError_t _err = (call);                                                         \
        if (_err != hipSuccess)                                                           \
            return fx_fail((e), FX_EHIP, std::string(#call) + ": " + hipGetErrorString(_err)); \
    } while (0)

int fx_scratch(fx_engine* e, int slot, size_t bytes, void** out);
int fx_pinned(fx_engine* e, int slot, size_t bytes, void** out);
// Device memory that is all zeros whenever no kernel is running: meeting points of the multi-workgroup small-batch forms
// (maxima through atomicMax, arrival tickets).  Zeroed when (re)allocated; the last workgroup to use an entry resets it.
int fx_zero_pool(fx_engine* e, size_t bytes, void** out);
int fx_upload_lut(fx_engine* e, const uint8_t lut[256]);
// device timeline buffer for the launch about to be enqueued (zeroed), or nullptr when the "trace" option is off
#define FX_TRACE_BYTES ((size_t)1024 * 16 * 16 * 8)
int fx_trace_buffer(fx_engine* e, unsigned long long** out);

// Deferred error words: mapped pinned HOST memory (read by the host right after the stream sync, no copy).  One 32-bit word PER
// error bit (word log2(bit) of the 64-byte line), so raising one is an idempotent system-scope store that cannot wipe another
// kernel's different error -- no PCIe atomics needed.  The host reads them OR-ed (fx_err_read).
#define FX_ERR_WORDS 3
inline unsigned fx_err_read(const unsigned* h_err) {
    unsigned v = 0;
    for (int i = 0; i < FX_ERR_WORDS; ++i) v |= reinterpret_cast<const volatile unsigned*>(h_err)[i];
    return v;
}
inline void fx_err_clear(unsigned* h_err) {
    for (int i = 0; i < FX_ERR_WORDS; ++i) reinterpret_cast<volatile unsigned*>(h_err)[i] = 0;
}
#if defined(__HIPCC__)
__device__ __forceinline__ void fx_raise(unsigned* err, unsigned bit) {
    __hip_atomic_store(err + (31 - __clz(bit)), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Relay reader: wave-uniform wait for flags[tile] == seq (device scope; the tile's bytes were stored before the flag, both past the
// caches).  Gives up like fx_rows_wait.
__device__ __forceinline__ void fx_relay_wait(const unsigned* flag, unsigned seq, unsigned* err) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        unsigned v = 0;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v == seq) break;
        if (wall_clock64() - t0 > 25000000ull) { if ((threadIdx.x & 63) == 0) fx_raise(err, FX_ERR_STARVED); break; }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Wave-uniform: returns when the rows of stage `stage` are in the staging area.  `known` = stages this wave has already seen
// published (the words are only read again for a later stage: once the host has finished packing, one poll settles the rest of the
// kernel).  The poll is a system-scope load; NO acquire fence or cache invalidate follows it (a workgroup-scope fence only keeps the
// compiler from hoisting the row loads): the staging area is coherent host memory, which no device cache level holds, and it is
// tile-pitched, so a line fetched for one tile never carries a neighbour's rows from before they were packed (FxRowsReady; an
// agent- or system-scope acquire here costs a launch ~85 us, profiles/r5_launch_first.log).  A host that dies
// mid-call must not hang the device: after 0.25 s the wave raises FX_ERR_STARVED and goes on (the host redoes the call).
__device__ __forceinline__ void fx_rows_wait(const FxRowsReady& r, int stage, int& known, unsigned* err) {
    if (stage < known) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        int have = 0x7FFFFFFF;
        if (lane < r.lanes) have = (int)(__hip_atomic_load(r.words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - r.base);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { const int other = __shfl_xor(have, o); have = other < have ? other : have; }   // (lanes 0 .. 15 hold every word)
        have = __builtin_amdgcn_readfirstlane(have);
        known = (have < 0 || have > r.Q) ? 0 : have;             // (a value of an earlier call lies below this call's base; nothing of this call lies above base + Q)
        if (stage < known) break;
        if (wall_clock64() - t0 > 25000000ull) { if (lane == 0) fx_raise(err, FX_ERR_STARVED); known = r.Q; break; }
        __builtin_amdgcn_s_sleep(64);                            // (~1.7 us: a thousand waiting waves must not flood the line the host stores into)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       // (the rows are read after the poll; no invalidate: see FxRowsReady)
}
#endif

// ---------------------------------------------------------------- kernel launchers
// (defined in the .hip files; all enqueue on e->stream and return an fx_status)
int fx_launch_score_generic(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                            int64_t N, float* d_out_NM, int Mtot, int m_off);
// returns FX_EUNSUPPORTED if no MFMA instantiation matches (caller falls back to generic)
int fx_launch_score_cnn_mfma(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                             int64_t N, float* d_out_NM, int Mtot, int m_off);
int fx_launch_cnn_pair_conv(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, void* d_pool);
int fx_launch_score_cnn_split(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                              int64_t N, float* d_out_NM, int Mtot, int m_off);
// small launches of the canonical 4-letter CNN, L <= 16 (TF-binding, RNA 14): one tile shared by a wave quad (score_cnn_quad.hip)
int fx_launch_score_cnn_quad(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                             float* d_out_NM, int Mtot, int m_off);
// the resident form: one workgroup per model on `stream`, answering `d_mail`; *cap = sequences per request it serves.
// FX_EUNSUPPORTED when the shape has no resident instantiation.
// tell a resident generation to leave (its workgroups poll the flag between requests); the next eligible calls start a new one
// host stores into device memory go through the write-combining BAR mapping: push them out
inline void fx_bar_fence() {
#if !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_sfence();
#endif
}
inline void fx_server_stop(fx_engine* e) {
    auto& sv = e->server;
    if (!sv.running) return;
    // "leave" is a request word of its own (one load per poll on the device side): a value no request can have -- sequence
    // number 0 is never used -- in both copies of the word; `stop` (read once per request loop) says the same for good measure
    sv.in->stop = 1;
    sv.in->req_wide = FX_SERVE_LEAVE;
    sv.in->req = FX_SERVE_LEAVE;
    fx_bar_fence();
    sv.running = false;
}
void fx_lp_disarm(fx_engine* e);      // a pre-launched instance of the layer-parallel form leaves (fx_resident.hip)
int fx_launch_score_cnn_quad_server(fx_engine* e, fx_model* const* models, int M, int m_off, int tiles, hipStream_t stream,
                                    FxMailIn* d_in, FxMailOut* d_out, unsigned long long idle_ticks, unsigned long long life_ticks,
                                    int want_quads, int* quads_out);
int fx_launch_score_dense_small_server(fx_engine* e, fx_model* const* models, int M, int m_off, int tiles, hipStream_t stream,
                                       FxMailIn* d_in, FxMailOut* d_out, unsigned long long idle_ticks, unsigned long long life_ticks);
// The resident workgroups' wait for the next request (thread 0 of a workgroup): returns the new request word, or `last` with
// *leave = 1 when told to stop, idle for too long or too old.  `fast` slots spin on `req`; the others poll `req_wide` (the same
// word, written first) with `sleep_n` x 64 clocks between polls, so that they do not queue in front of the fast slots' line.
#if defined(__HIPCC__)
// A streamed request's tile: wait until the host has packed `need` rows.  false = abandon the request (the host said LEAVE, or
// nothing came for 2 s: the tile must NOT be answered -- the host falls back to a launch when an answer stays away).
// Called by whole waves (every lane polls the same word: one request per wave and poll).
__device__ __forceinline__ bool fx_server_rows_ready(const FxMailIn* in, unsigned long long req, long long need) {
    if (!(req & FX_SERVE_STREAM)) return true;
    const unsigned long long tag = req >> 16, t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long r = __hip_atomic_load(&in->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((r >> 16) == tag && (long long)(r & 0xFFFFull) >= need) return true;
        if ((spins & 63u) == 63u) {
            if (__hip_atomic_load(&in->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == FX_SERVE_LEAVE) return false;
            if (wall_clock64() - t0 > 200000000ull) return false;
        }
    }
}
// The same wait by ALL lanes of a wave, for the slot that answers tile 0 (the only tile of a tiny request): every poll reads the
// whole request line -- lane l dword l & 15, one 64-byte read -- and the lanes keep their dword in *payload: when the word that
// comes back carries FX_SERVE_TINY, lanes 2 .. 13 hold the request's bytes.  A line whose two copies of the word differ (a read
// that overtook half of the host's write) is read again.
__device__ __forceinline__ unsigned long long fx_server_wait_line(const FxMailIn* in, unsigned long long last, unsigned long long seen,
                                                                  unsigned long long start, unsigned long long idle_ticks,
                                                                  unsigned long long life_ticks, int lane, int* leave, unsigned* payload) {
    *leave = 0;
    const unsigned* line = reinterpret_cast<const unsigned*>(in) + (lane & 15);
    for (;;) {
        const unsigned v = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long r = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 0) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 1) << 32);
        if (r == FX_SERVE_LEAVE) { *leave = 1; return last; }
        if (r != last) {
            const unsigned long long tail = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 14) |
                                            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 15) << 32);
            if (!(r & FX_SERVE_TINY) || tail == r) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                *payload = v;
                return r;
            }
        }
        const unsigned long long now = wall_clock64();
        if (now - seen > (last ? idle_ticks : 64 * idle_ticks) || now - start > life_ticks) { *leave = 1; return last; }
    }
}
__device__ __forceinline__ unsigned long long fx_server_wait(const FxMailIn* in, unsigned long long last, unsigned long long seen,
                                                             unsigned long long start, unsigned long long idle_ticks,
                                                             unsigned long long life_ticks, bool fast, int sleep_n, int* leave) {
    *leave = 0;
    const unsigned long long* word = fast ? &in->req : &in->req_wide;
    for (;;) {
        // relaxed polls, ONE acquire fence when the word changes: an acquire load is an L2 invalidate per iteration, and
        // 240 workgroups invalidating their L2s in a loop delayed everybody (tools/probes/mailbox_probe4.hip: request round
        // trip of 3 workgroups beside 237 idle ones 6.8 us with acquire polls, 4.7 us with relaxed ones; 3.5 us alone)
        const unsigned long long r = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r == FX_SERVE_LEAVE) { *leave = 1; return last; }            // (told to leave: the word itself says so -- ONE load per poll)
        if (r != last) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return r; }
        const unsigned long long now = wall_clock64();
        if (now - seen > (last ? idle_ticks : 64 * idle_ticks) ||       /* (a generation waits longer for its first request) */
            now - start > life_ticks) { *leave = 1; return last; }
        if (!fast)
            for (int k = 0; k < sleep_n; ++k) __builtin_amdgcn_s_sleep(1);
    }
}
#endif
int fx_launch_score_cnn_pair(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                             int64_t N, float* d_out_NM, int Mtot, int m_off);
// small launches of the MLP: a tile's output tiles dealt to the waves of a workgroup (score_dense_small.hip)
int fx_launch_score_mlp_small(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N,
                              float* d_out_NM, int Mtot, int m_off);
int fx_mlp_first_layer_form(fx_engine* e, const FxShape& s, const FxPackLayout& lay);
// MLP whose first-layer rows do not fit LDS beside the rest of the image and that takes its first layer position-major at batch size (option mlp_l1_pos, score_dense_l1.h)
bool fx_mlp_l1_pos_applies(const fx_engine* e, const FxShape& s, const FxPackLayout& lay);
int fx_launch_score_dense_mfma(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii,
                               int64_t N, float* d_out_NM, int Mtot, int m_off);
int fx_launch_mfma_probe(fx_engine* e, const float* d_a, const float* d_b, const float* d_c, float* d_d);
int fx_launch_encode_onehot(fx_engine* e, const uint8_t* d_ascii, int64_t N, int L, int A, float* d_out);
int fx_launch_ensemble_mean_planar(fx_engine* e, const float* d_planes, int64_t N, int M, int64_t stride, float* d_out32);
int fx_launch_ensemble_reduce(fx_engine* e, const float* d_scores, int64_t N, int M,
                              const double* d_weights, float* d_out32, double* d_out64);
int fx_launch_argmax_decode(fx_engine* e, const double* d_onehot, int64_t rows, int A,
                            const uint8_t* d_alphabet, uint8_t* d_out);
int fx_launch_nam_combine(fx_engine* e, int64_t Q, const double* d_signal, const double* d_noise,
                          const int32_t* d_dist, const double* d_alpha, int n_tab, double* d_out);
int fx_launch_min_dist(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache,
                       int64_t C, int L, unsigned long long* d_keys);
int fx_launch_distances_bounded(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C, int L,
                                int K, uint8_t* d_out);
int fx_launch_distances(fx_engine* e, int mode, const uint8_t* d_q, int64_t Q, const uint8_t* d_cache, int64_t C,
                        int L, uint8_t* d_out);
int fx_launch_additive_sum(fx_engine* e, const double* d_table, int L, int ncol, const uint8_t* d_ascii, int64_t N,
                           double* d_out);
int fx_launch_nam_table_blend(fx_engine* e, int64_t Q, const uint8_t* d_queries, const uint8_t* d_keys, const int64_t* d_arg,
                              const int32_t* d_dist, const double* d_table, int64_t len, int L, int bits, const double* d_E,
                              const double* d_alpha, int n_tab, double* d_out, int32_t* d_flags);
int fx_launch_table_lookup(fx_engine* e, const double* d_table, int64_t len, const uint8_t* d_ascii, int64_t N,
                           int L, int bits, double* d_out);
int fx_launch_min_dist_finish(fx_engine* e, const unsigned long long* d_keys, int64_t Q, int64_t C,
                              int32_t* d_dist, int64_t* d_arg);
