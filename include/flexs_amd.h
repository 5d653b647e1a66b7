/*
 * flexs_amd.h -- C ABI of libflexs_amd.so, the MI355X (gfx950) scoring engine
 * behind the FLEXS `Model.get_fitness` hot path.
 *
 * The reference (samsinai/FLEXS v0.2.7) is pure Python and has no FFI of its
 * own; the boundary below is what a ctypes binding inside the reference's
 * `KerasModel._fitness_function` / `Ensemble._fitness_function` /
 * `NoisyAbstractModel._fitness_function` would call.
 * Each entry point cites the reference code (paths relative to the FLEXS
 * repository root) it replaces.
 *
 * Conventions
 *   - plain C types only; every function returns an fx_status (0 = FX_OK,
 *     negative = error) unless stated; the process is never aborted.
 *   - "host" entry points take host pointers, are synchronous, and own all
 *     staging (results are valid on return).
 *   - "_dev" entry points take DEVICE pointers (HBM-resident buffers, e.g.
 *     torch tensor .data_ptr()), enqueue on the engine's stream and return
 *     immediately; call fx_engine_sync() to wait and to collect deferred
 *     errors (unknown alphabet character).
 *   - the caller owns every buffer it passes; the engine owns device scratch,
 *     its stream (unless one is lent with fx_engine_set_stream) and the packed
 *     weights behind fx_model handles.
 *   - an engine handle is not thread-safe: one per host thread / per GPU.
 */
#ifndef FLEXS_AMD_H
#define FLEXS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_VERSION 100 /* 0.1.0 */

typedef enum fx_status {
    FX_OK = 0,
    FX_EINVAL = -1,       /* bad argument (null pointer, negative size, ...)          */
    FX_ESHAPE = -2,       /* shape mismatch: len(seq) != seq_len, L < kernel_size ... */
    FX_EBADCHAR = -3,     /* character not in the alphabet (str.index ValueError,
                             flexs/utils/sequence_utils.py:46)                        */
    FX_ENODEV = -4,       /* no usable gfx950 device                                  */
    FX_EHIP = -5,         /* HIP runtime error (text in fx_last_error)                */
    FX_ENOMEM = -6,
    FX_EUNSUPPORTED = -7, /* configuration outside what the kernels implement         */
    FX_ESTATE = -8        /* e.g. weights never set                                   */
} fx_status;

typedef enum fx_model_kind {
    FX_CNN = 0, /* flexs/baselines/models/cnn.py:23-54                    */
    FX_MLP = 1, /* flexs/baselines/models/mlp.py:21-31                    */
    FX_GE = 2   /* flexs/baselines/models/global_epistasis_model.py:26-36 */
} fx_model_kind;

typedef enum fx_dist_mode {
    FX_LEVENSHTEIN = 0, /* editdistance.eval, noisy_abstract_model.py:51 (parity default) */
    FX_HAMMING = 1      /* BASELINE.json north-star wording; upper bound of the above     */
} fx_dist_mode;

typedef struct fx_engine fx_engine; /* one GPU: stream, scratch, deferred-error word   */
typedef struct fx_model fx_model;   /* one surrogate: shape + packed device weights    */
typedef struct fx_cache fx_cache;   /* device-resident NoisyAbstractModel cache keys   */
typedef struct fx_table fx_table;   /* device-resident k-mer -> fitness table landscape */

/* ------------------------------------------------------------------ library */
int fx_version(void);
const char *fx_status_name(int status);
int fx_device_count(void); /* >= 0, or negative fx_status */

/* ------------------------------------------------------------------- engine */
int fx_engine_create(int device, fx_engine **out);
int fx_engine_destroy(fx_engine *e);
/* Lend an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL restores the engine's own stream. */
int fx_engine_set_stream(fx_engine *e, void *hip_stream);
/* Wait for the stream; returns FX_EBADCHAR if any _dev call since the last
 * sync met a character outside its alphabet. */
int fx_engine_sync(fx_engine *e);
/* Stream-ordered: *d_dst (device memory) = the engine's deferred error bits (1 = character outside the alphabet) as a float,
 * covering every _dev call enqueued before it; the bits stay set for fx_engine_sync.  Multi-GPU callers place it in the block
 * they all-gather, so an error on any rank reaches every rank with the scores -- the path's ONE collective
 * (flexs/ensemble.py:54-59 sharded). */
int fx_engine_error_word_dev(fx_engine *e, float *d_dst);
const char *fx_last_error(fx_engine *e);
/* Engine options (string key -> int64).  Unknown key -> FX_EINVAL.  The defaults are what the measurements under
 * profiles/ selected; an integrator normally sets none of them.
 *
 *   key               default  meaning
 *   ----------------  -------  ----------------------------------------------------------------------------------
 *   grid_blocks       0        workgroups of the persistent scoring kernels; 0 = one per CU.  Multi-GPU callers leave
 *                              a few CUs to RCCL's channel kernels (bench.py: num_cus - 4).
 *   num_cus           (read)   compute units of the device (get only).
 *   force_generic     0        1 = score with the shape-agnostic VALU kernels instead of the MFMA ones: an independent
 *                              on-device cross-check (~80x slower), used by the tests.
 *   poison_outputs    0        1 = NaN-fill score buffers before every launch, so that an element no kernel wrote is
 *                              caught (test aid).
 *   trace             0        1 = the MFMA scoring kernels stamp an in-kernel timeline (fx_debug_trace_read; needs the
 *                              `make trace` build).
 *   zero_copy_bytes   262144   host calls whose input + output fit in this many bytes read / write mapped pinned
 *                              host memory directly (explorer-size calls: no copy enqueues) -- fx_score, and since
 *                              round 3 fx_decode_score, fx_min_dist / fx_cache_min_dist, fx_nam_combine, fx_table_*.
 *   zero_copy_mode    -1       larger host calls: -1 = decide per call (fx_plan_host_call), 0 = always copy,
 *                              1 = always zero-copy.
 *   launch_first      1        1 = fx_score_begin_staged is offered (big zero-copy calls of callers that marshal strings:
 *                              kernels enqueued before the strings are packed); 0 = it answers FX_EUNSUPPORTED.  Needs a
 *                              large BAR, as serve_small does.  Read-only companions: launch_first_calls, launch_first_redone.
 *   launch_relay      1        1 = host calls of dense ensembles whose plan says "copy" (fx_plan_host_call) run without an
 *                              upload in front of the launch: member 0's workgroups read the staging area and pass every tile
 *                              on to the other members through device memory (fx_score, and fx_score_begin_staged, which then
 *                              takes such calls too).  Read-only companion: launch_relay_calls.
 *   serve_small       1        1 = explorer-size fx_score calls of canonical 4-letter CNNs (seq_len <= 16), MLPs and
 *                              GlobalEpistasis models -- one model, an ensemble, or a mix -- are answered by workgroups
 *                              that STAY on the device between calls (request and answer through mailboxes; no launch):
 *                              28 -> 10 us per 20-sequence call, same bits.  They start when the same model list calls
 *                              twice within serve_idle_us; new weights, training, ANY option change, a launch that fills
 *                              the chip, or engine destruction tell them to leave.  Needs a large BAR (the host stores
 *                              the request into device memory; verified once from /proc/self/maps + a read-back, never
 *                              by faulting); 0 = a launch per call.
 *   serve_wide        1        geometry of a resident generation.  1 = ADAPTIVE: narrow (<= 16 tile slots per member on a
 *                              third of the CUs, <= 256 sequences per request: round 3's form) for callers that only ask
 *                              for a few sequences; WIDE -- (num_cus - serve_reserve_cus) / members tile slots, a slot
 *                              walking the tiles slot, slot + T, ... of a request: up to 4096 sequences / 64 KiB per
 *                              request, 2001 8-mers in 26 us instead of 39 -- as soon as the caller asks for more than
 *                              256 sequences twice within 2 ms (240 resident workgroups cost every explorer-size call
 *                              ~1.2 us, hence not always).  2 = always wide, 0 = always narrow (A/B).
 *   serve_reserve_cus 16       CUs a wide generation leaves without a resident workgroup.
 *   serve_poll_sleep  8        s_sleep units between polls of the tile slots beyond the first few of a wide generation
 *                              (they poll a copy of the request word on a line of its own).
 *   serve_fence       0        1 = round 3's system fence after every tile's answers (A/B; the answers are system-scope
 *                              stores, which write through by themselves).
 *   host_mean_below   256      launched mean-only host calls of at most this many sequences (the protein CNN's explorer-size
 *                              calls): member planes straight to pinned host memory, np.mean's order on the host, no mean
 *                              launch; same bits.  0 = the mean kernel.  call_prof_0 .. _3 (read): that call's timeline, ns
 *                              (prepared, launched, synchronised, mean taken).
 *   lp_prelaunch      1        1 = after an explorer-size host call of a protein CNN ensemble was answered by the layer-parallel
 *                              form, the NEXT instance of that call (same members, batch size, output form) is enqueued at
 *                              once: it fills its weights and waits up to serve_idle_us for its request word in a mailbox
 *                              the host stores into through the BAR -- a caller that is back within that window (CMA-ES,
 *                              DyNA-PPO) pays neither launch latency nor weight fill: 37.8 -> 31.9 us per call.  The waiting
 *                              instance holds most CUs; a call of another shape, training, any option change tell it to
 *                              leave at once, work enqueued on the engine's stream by anything else waits for its idle exit
 *                              (<= serve_idle_us).  0 = a launch per call.  lp_armed_served (read): calls answered so.
 *   done_flag         1        1 = a launched small host call whose last kernel can tell when its last result is written (the
 *                              layer-parallel protein CNN form) is waited for by polling a completion word in pinned host
 *                              memory instead of hipStreamSynchronize: ~4 us of a 46 us call; 0 = always the stream.
 *   serve_tiny        1        1 = a request of at most 48 sequence bytes (one to six 8-mers) carries them in the request word's
 *                              own 64-byte line; the workgroup of tile 0 reads the whole line per poll and needs no second
 *                              read of device memory for the bytes (~0.2 us of a 9.4 us call).  0 = always the byte area.
 *   dist_bounded      1        fx_cache_density with radius 1 .. 3: 1 = distances up to the radius by a banded kernel that
 *                              leaves a pair once it is known to be farther; 0 = the exact distance matrix.  Same densities.
 *   dist_stage        1        edit-distance kernels, small launches (fewer than 512 blocks' worth): 1 = a block copies its 256
 *                              cache rows (<= 160 bytes each) to LDS and the recurrence reads them there; 0 = from global memory.
 *   serve_quads       1        wide generation, 4-letter CNN with seq_len <= 8: tiles a resident workgroup answers side by
 *                              side.  3 (the launched form's three quads) is in the A/B build only: slower once requests
 *                              are streamed (fx_score_stream_*).
 *   serve_idle_us     500      ... calls further apart than this are launched; the workgroups leave by themselves after
 *                              twice this long without a request.  A device-wide synchronize (hipDeviceSynchronize)
 *                              issued right after a small call waits for that.
 *   server_calls, server_starts, server_fallbacks, server_resident, server_wide, server_slots, server_streamed   (read)
 *                              bookkeeping of the resident form; server_prof_0 .. _7 (read): the last served call's
 *                              timeline, ns since it entered the library (admitted, posted, first answer, collected,
 *                              outputs written, member planes 0 - 2 done).
 *   ab_build          (read)   1 = this library is the A/B build (`make -C flexs_amd/csrc ab`): the kernel forms that were
 *                              measured and LOST are compiled in; the production library refuses the option values that
 *                              select them (FX_EUNSUPPORTED).
 *   cnn_lp            1        1 = small batches of the canonical protein CNN run layer-parallel over the chip (conv2
 *                              outputs through device memory, one grid barrier) instead of position segments with
 *                              recomputed halos: a 237-residue call 64 -> 24 us, same bits.  0 = the segmented form.
 *   train_canon       1        fx_train_fit: canonical shapes run the step instantiated with compile-time dimensions
 *                              (k_train_fb 47.8 -> 29.8 us with the padded LDS rows, same bits); 0 = the shape-agnostic
 *                              code for everything.
 *   train_swizzle     3        fx_train_fit, CNNs whose weights do not fit LDS (the 20-letter alphabets).  3 = any CNN with 32
 *                              filters and at most 16 M tiles per slice (one row per slice up to 260 residues) runs the F = 32
 *                              form: fragment-order rows and staged kernels (every MFMA operand one ds_read_b128), paired
 *                              output tiles, tap groups carried in registers across phases, sliding-window weight gradient,
 *                              a canonical instantiation for CNN(32, 100, kernel 5) on 20 letters (3 x CNN at 237 residues on
 *                              500 sequences: 43.9 -> 17.9 ms).  2 / 1 = round 4's forms for 226 ... 239 positions (staged
 *                              conv kernels over rotated rows / rotated rows only: 31.6 / 42.1 ms), 0 = the plain step.  Same
 *                              bits in every form (GPU tests), each held to the float64 restatement at 90 ... 300 residues.
 *   train_persistent  0        fx_train_fit: 1 = the whole fit as ONE launch (member barriers in device memory, Adam by the
 *                              same workgroups; needs all workgroups co-resident).  Same bits, no faster: off.
 *   train_rows        0        fx_train_fit: mini-batch rows per forward+backward workgroup; 0 = automatic (depends on
 *                              the member's own shape and batch size only: a fit is bit-reproducible whatever it is
 *                              trained next to).
 *   train_threads     0        fx_train_fit: threads per forward+backward workgroup (256 / 512 / 1024); 0 = 1024.
 *   train_lds         2        fx_train_fit: 2 = activations, gradients and weights of a slice live in LDS when they
 *                              fit, 1 = activations and gradients only, 0 = global memory.
 *
 * Kernel-form selectors -- every one chooses between forms that give the SAME BITS (tested), so they are speed knobs
 * only; the defaults are the measured winners and the losers stay selectable as the A/B baseline of the profiles:
 *   cnn_variant, cnn_big_units, cnn_pair, cnn_seg, cnn_seg_multi, cnn_pair_seg, cnn_pair_seg4, cnn_quad, quad_rotate   (CNN launch forms)
 *   dense_small, dense_slab, dense_waves, dense_few_waves_below, dense_pipe, dense_coop, mlp_pair, ge_bytetab   (MLP / GE forms)
 *   stage_bytes, stage_fill, dma_fill, wave_prio                                                             (staging / scheduling)
 * A/B build only (the production library answers FX_EUNSUPPORTED): cnn_conv1_mfma, mlp_l1_mfma (one-hot first layers on MFMA
 * instead of the LDS gather), dense_pipe, fuse_mean, chunk_overlap, dense_waves = 8, dense_few_waves_below, cnn_pair = 0,
 * cnn_variant 2 / 3 / 5 / 6, train_split, serve_quads = 3 -- each measured slower than the default.
 * What each does, its values and the measurement that decided it: flexs_amd/csrc/OPTIONS.md. */
int fx_engine_set_option(fx_engine *e, const char *key, int64_t value);
int fx_engine_get_option(fx_engine *e, const char *key, int64_t *value);
/* Engine-side counters since creation (or the last reset): out9 = { host scoring calls, device-buffer scoring launches,
 * sequences scored, forwards (sequences x members), bytes copied host -> device, bytes copied device -> host (copy path
 * only), host calls served zero-copy, (query, cache entry) distance evaluations, training steps x members }.
 * reset != 0 zeroes them after reading. */
int fx_engine_counters(fx_engine *e, int64_t *out9, int reset);
/* hipEvent pair on the engine's stream: start..stop brackets whatever was
 * enqueued in between; stop synchronises and returns elapsed milliseconds. */
int fx_timer_start(fx_engine *e);
int fx_timer_stop(fx_engine *e, float *elapsed_ms);

/* -------------------------------------------------------------------- model */
/* Replaces the tf.keras.Sequential built in cnn.py:23-54 / mlp.py:21-31 /
 * global_epistasis_model.py:26-36.  L = seq_len, A = len(alphabet),
 * F = num_filters (CNN only), H = hidden_size, K = kernel_size (CNN only).
 * FX_ESHAPE if kind == FX_CNN and L < K (Keras 'valid' conv raises). */
int fx_model_create(fx_engine *e, int kind, int L, int A, int F, int H, int K, fx_model **out);
int fx_model_destroy(fx_model *m);
int64_t fx_model_num_params(const fx_model *m);
/* Weight blob = the arrays of keras `model.get_weights()` flattened (C order)
 * and concatenated: CNN  conv1 kernel[K][A][F], bias[F], conv2 kernel[K][F][F],
 * bias[F], conv3 kernel[A-1][F][F], bias[F], dense[F][H], bias[H], dense[H][H],
 * bias[H], dense[H][1], bias[1];  MLP  dense[L*A][H], b, dense[H][H], b,
 * dense[H][H], b, dense[H][1], b;  GE  dense[L*A][1], b[1], dense[1][H], b,
 * dense[H][H], b, dense[H][1], b[1].  Called once per explorer round after
 * `train` (flexs/explorer.py:157-160). */
int fx_model_set_weights(fx_model *m, const float *blob, int64_t n);
int fx_model_get_weights(const fx_model *m, float *blob, int64_t n);

/* ------------------------------------------------------------------ scoring */
/* fx_score in pieces, for callers whose input has to be marshalled first (Python strings): fill the
 * returned staging area (N x L bytes, pinned) chunk by chunk and submit each chunk of rows as soon as
 * it is ready -- host marshalling of chunk k+1 then overlaps the transfer and scoring of chunk k.
 * fx_score_finish waits, reports a bad character (FX_EBADCHAR) and copies the results out.  One such
 * call per engine at a time; chunks should start at multiples of 16 rows. */
/* How the engine would move the data of a host call of N sequences: *zero_copy = 1 when the kernels will read the
 * sequences from / write the scores to mapped pinned memory directly (no copies; chosen when M x N x L bytes of PCIe
 * reads hide behind the kernels), *pieces = how many pieces a caller that marshals its input (Python strings) should
 * submit (1 = a plain fx_score on the staging area). */
int fx_plan_host_call(fx_engine *e, fx_model *const *models, int M, int64_t N, int L, int *zero_copy, int *pieces);
int fx_score_begin(fx_engine *e, fx_model *const *models, int M, int64_t N, int L,
                   const uint8_t lut[256], int want_nm, int want_mean, void **staging);
int fx_score_submit(fx_engine *e, int64_t row0, int64_t rows);
int fx_score_finish(fx_engine *e, float *out_NM, float *out_mean);
/* The same call launched FIRST and packed behind (replaces nothing in the reference -- its `get_fitness` encodes, then
 * predicts, keras_model.py:69-79; this is how the encoding of a big list of str hides behind the GPU's work): the kernels
 * are enqueued by this function and read the staging area tile by tile as the caller fills it.  The staging area of such a
 * call is TILE-PITCHED: the (up to) 16 rows of tile t, L bytes each, start at t * *tile_pitch (16 L rounded up to whole
 * 128-byte lines).  The caller packs in STAGES -- stage j = the tiles t with t % *stages == j -- split over `lanes` packing
 * threads (lane l takes the l-th of `lanes` equal parts of each stage's tile list), and after finishing its part of stage j
 * lane l stores *base + j + 1 into words[l] (device memory mapped into the host: a store fence in front of the store and
 * one behind it), then calls fx_score_finish as for the call in pieces -- after fx_score_abandon if it could not pack
 * every row (finish then only waits for the kernels; the results are meaningless).  FX_EUNSUPPORTED = not for this call
 * (nothing enqueued, no call in flight): use fx_score_begin or fx_score.  Engine option launch_first = 0 turns it off. */
int fx_score_begin_staged(fx_engine *e, fx_model *const *models, int M, int64_t N, int L,
                          const uint8_t lut[256], int want_nm, int want_mean, int lanes,
                          void **staging, void **words, unsigned *base, int *stages, int *tile_pitch,
                          void *results, int64_t results_bytes);
int fx_score_abandon(fx_engine *e);
/* `results` of fx_score_begin_staged (nullable; `results_bytes` its size): a buffer of fx_result_alloc of at least
 * (want_nm ? 4 N M : 0) + (want_mean ? 4 N : 0) bytes that the kernels write the (N, M) matrix and then the mean INTO -- fx_score_finish then
 * copies nothing (its out_NM / out_mean are ignored) and the caller wraps the buffer (NumPy: an array whose base owns the
 * buffer and gives it back to a pool when the last view dies).  Pinned, GPU-mapped host memory; free with fx_result_free
 * once nothing refers to it. */
int fx_result_alloc(fx_engine *e, int64_t bytes, void **host);
int fx_result_free(fx_engine *e, void *host);

/* The engine's pinned, GPU-mapped input staging area, grown to at least `bytes`.  A caller that
 * marshals its strings straight into it (instead of into pageable memory) and then passes the
 * same pointer to fx_score saves that call's host-to-staging copy.  The pointer is valid until
 * the next fx_staging_input / fx_score on this engine asks for more. */
int fx_staging_input(fx_engine *e, int64_t bytes, void **host);

/* KerasModel._fitness_function (keras_model.py:69-79) for M models that share
 * (L, A) + Ensemble._fitness_function (ensemble.py:54-59) fused:
 *   ascii  N x L bytes, row-major, one sequence per row (no terminators)
 *   lut    256 entries: byte -> alphabet index, 0xFF = not in alphabet
 *   out_NM nullable, N x M float32, column m = models[m] (np.stack(axis=1))
 *   out_mean nullable, N float32 = np.mean(out_NM, axis=1) in NumPy's own
 *          float32 summation order (bit-exact w.r.t. the stacked matrix)
 * Scores are float32 with np.nan_to_num applied (keras_model.py:77). */
int fx_score(fx_engine *e, fx_model *const *models, int M, const uint8_t *ascii, int64_t N, int L,
             const uint8_t lut[256], float *out_NM, float *out_mean);
int fx_score_dev(fx_engine *e, fx_model *const *models, int M, const uint8_t *d_ascii, int64_t N,
                 int L, const uint8_t lut[256], float *d_out_NM, float *d_out_mean);
/* fx_score for a caller that still has to PACK its rows (a Python list of str, keras_model.py:69-79's argument): streamed
 * over the resident form (round 4).  _begin posts the request for N sequences and hands out the mailbox's byte area (device
 * memory behind the BAR: write it front to back, never read it); the caller packs rows into it in order and reports progress
 * with _rows (total rows packed so far, every few hundred rows -- a resident workgroup starts a 16-row tile as soon as those
 * rows are reported); _end(ok = 1) reports the last rows, collects the answers and writes out_NM / out_mean as fx_score
 * does.  The packing of a 2001-string call then runs beside the first tiles instead of in front of them.
 *   _begin: FX_OK, or FX_EUNSUPPORTED when the resident form does not take this call now (no generation running for these
 *           members, too many sequences, first request of a generation ...): pack into own memory and call fx_score.
 *   _end:   ok = 0: the caller could not pack (not a str, ragged ...): the request is abandoned, FX_OK.
 *           FX_EUNSUPPORTED: the generation went away before it answered: pack into own memory and call fx_score.
 * Same results, bit for bit, as fx_score on the packed rows. */
int fx_score_stream_begin(fx_engine *e, fx_model *const *models, int M, int64_t N, int L, const uint8_t lut[256], uint8_t **rows);
int fx_score_stream_rows(fx_engine *e, int64_t rows_packed);
int fx_score_stream_end(fx_engine *e, int ok, float *out_NM, float *out_mean);
/* The mean-only device path in two calls, with the intermediate in the caller's hands: the M members' scores as
 * member-major PLANES (`d_planes[m * stride + n]`, stride >= N, a multiple of 4, base 16-byte aligned) -- a work
 * unit's 16 scores are then one contiguous 64-byte store instead of 16 stores 4*M bytes apart as in the (N, M)
 * matrix of ensemble.py:55-57 -- then np.mean over the planes (M <= 16, NumPy summation order).  fx_score_dev
 * uses the same layout internally when it is asked for the mean only. */
int fx_score_planes_dev(fx_engine *e, fx_model *const *models, int M, const uint8_t *d_ascii, int64_t N,
                        int L, const uint8_t lut[256], float *d_planes, int64_t stride);
int fx_ensemble_mean_planes_dev(fx_engine *e, const float *d_planes, int64_t N, int M, int64_t stride,
                                float *d_out_mean);
/* Both halves as ONE call (ensemble.py:54-59 with the default np.mean): the planes hold the members' scores and
 * `d_out_mean` their NumPy-order mean afterwards.  Where the scoring kernel can take the mean itself -- the batch
 * form of the 4-letter CNN: the last member to finish a 16-sequence tile averages it -- no second kernel is
 * launched (engine option "fuse_mean_batch", default 1); same bits either way.  M <= 16. */
int fx_score_mean_planes_dev(fx_engine *e, fx_model *const *models, int M, const uint8_t *d_ascii, int64_t N,
                             int L, const uint8_t lut[256], float *d_planes, int64_t stride, float *d_out_mean);


/* string_to_one_hot over a batch (sequence_utils.py:32-47 + keras_model.py:70-75):
 * N x L bytes -> N x L x A float32 0/1.  Stand-alone, HBM-bound. */
int fx_encode_onehot(fx_engine *e, const uint8_t *ascii, int64_t N, int L, const uint8_t lut[256],
                     int A, float *one_hot);
int fx_encode_onehot_dev(fx_engine *e, const uint8_t *d_ascii, int64_t N, int L,
                         const uint8_t lut[256], int A, float *d_one_hot);

/* Ensemble / AdaptiveEnsemble combine (ensemble.py:24,59;
 * adaptive_ensemble.py:54,102): scores N x M float32 ->
 *   weights == NULL : np.mean(scores, axis=1)      (NumPy pairwise order)
 *   weights != NULL : np.sum(weights * scores, axis=1) computed in float64
 *                     (weights float64[M], out64 float64[N]); out32 unused. */
int fx_ensemble_reduce(fx_engine *e, const float *scores_NM, int64_t N, int M,
                       const double *weights, float *out32, double *out64);
int fx_ensemble_reduce_dev(fx_engine *e, const float *d_scores_NM, int64_t N, int M,
                           const double *weights, float *d_out32, double *d_out64);

/* one_hot_to_string (sequence_utils.py:50-66; callers cmaes.py:61-67,
 * environments/dyna_ppo.py:144-147): P x L x A float64 -> P x L bytes
 * alphabet[argmax], first maximum wins (np.argmax rule, NaN = maximum). */
int fx_argmax_decode(fx_engine *e, const double *one_hot, int64_t P, int L, int A,
                     const uint8_t *alphabet, uint8_t *out_chars);

/* Population step of the explorers that optimise in one-hot space: CMAES
 * `_soln_to_string` + `objective_function` (flexs/baselines/explorers/cmaes.py:61-67, 83-93) and
 * the DyNA-PPO environment step (environments/dyna_ppo.py:144-163) decode a float (L, A) array
 * to a sequence and score it, one sequence per `get_fitness` call.  fx_decode_score does both
 * for P solutions in one device round trip: out_chars[p] = alphabet[argmax] (fx_argmax_decode
 * rule), scores as fx_score would return for those sequences.  A decoded character outside
 * the model's alphabet -> FX_EBADCHAR. */
int fx_decode_score(fx_engine *e, fx_model *const *models, int M, const double *one_hot, int64_t P,
                    int L, int A, const uint8_t *alphabet, const uint8_t lut[256],
                    uint8_t *out_chars, float *out_NM, float *out_mean);

/* ----------------------------------------------------- NoisyAbstractModel */
/* NoisyAbstractModel._get_min_distance (noisy_abstract_model.py:42-60) for Q
 * queries against C cache keys kept in insertion order: dist[i] = min edit
 * distance, argmin[i] = FIRST cache index attaining it (the reference's
 * early-exit-at-1 + strict-< loop).  C == 0 -> dist 0, argmin -1
 * (noisy_abstract_model.py:44-45).  Rows are L bytes; a sequence shorter than L is
 * NUL-padded on the right (`editdistance.eval` accepts two strings of any lengths,
 * noisy_abstract_model.py:51; no FLEXS alphabet contains NUL).  FX_HAMMING compares the
 * padded rows byte-wise.  L <= 768 (up to 12 64-bit words per column; the full-length AAV capsid is 735). */
int fx_min_dist(fx_engine *e, int mode, const uint8_t *queries, int64_t Q, const uint8_t *cache,
                int64_t C, int L, int32_t *dist, int64_t *argmin);
/* Device-resident, append-only cache (self.cache keys, noisy_abstract_model.py:40,67,99). */
int fx_cache_create(fx_engine *e, int L, fx_cache **out);
int fx_cache_destroy(fx_cache *c);
int64_t fx_cache_size(const fx_cache *c);
int fx_cache_append(fx_cache *c, const uint8_t *keys, int64_t n);
int fx_cache_min_dist(fx_cache *c, int mode, const uint8_t *queries, int64_t Q, int32_t *dist,
                      int64_t *argmin);
/* NoisyAbstractModel's whole batch on a table landscape in ONE device round trip (noisy_abstract_model.py:69-94 with
 * flexs/landscapes/tf_binding.py:43-44 as the landscape): nearest cached neighbour of every query (as fx_cache_min_dist;
 * rows of fx_cache L bytes), the table's value of the query and of its neighbour (as fx_table_lookup with `bits`, `lut`),
 * noise = neighbour value x E[q] (E: the caller's standard-exponential draws, query order -- what
 * np.random.exponential(scale = neighbour value) returns), out[q] = alpha_tab[d] signal + (1 - alpha_tab[d]) noise
 * (as fx_nam_combine).  flags[q]: bit 0 = query or neighbour missing from the table (KeyError in the reference),
 * bit 1 = negative neighbour value (the reference then draws from the cache, :90-91) -- in either case the caller
 * restores its RNG state and takes the general path.  An empty cache: distance 0, the query is its own neighbour.
 * append_keys (n_append rows, may be 0): keys that join the cache before the search, as by fx_cache_append -- the
 * sequences the previous batch cached -- in the same submission when they fit the staging area.  Q <= 32768. */
int fx_cache_nam_query(fx_cache *c, fx_table *t, int bits, const uint8_t lut[256], int mode, const uint8_t *append_keys,
                       int64_t n_append, const uint8_t *queries, int64_t Q, const double *E, const double *alpha_tab,
                       int n_tab, double *out, int32_t *dist, int64_t *argmin, int32_t *flags);
/* noisy_abstract_model.py:88-94 for Q uncached queries:
 *   out[i] = alpha_tab[d[i]] * signal[i] + (1 - alpha_tab[d[i]]) * noise[i]
 * with alpha_tab[k] = ss ** k built on the host (Python float pow) and
 * noise[i] = scale[i] * standard_exponential draw (host legacy NumPy RNG,
 * bit-identical to np.random.exponential(scale)). float64 throughout. */
int fx_nam_combine(fx_engine *e, int64_t Q, const double *signal, const double *noise,
                   const int32_t *d, const double *alpha_tab, int n_tab, double *out);

/* Dense distance matrix of Q queries against the device-resident keys: out[q*C + c] =
 * min(distance, 255) as uint8 (C = fx_cache_size).  Serves `sequence_density`
 * (flexs/baselines/explorers/environments/dyna_ppo.py:106-114,267-275: sum of
 * fitness/dist over all seen sequences with 0 < dist <= 2): the host filters by radius
 * and accumulates in insertion order, so the float sum stays bit-identical. */
int fx_cache_distances(fx_cache *c, int mode, const uint8_t *queries, int64_t Q, uint8_t *out_QxC);
/* `sequence_density` itself (dyna_ppo.py:106-114) for Q queries: density[q] = sum over the stored keys i, in insertion order, of
 * fitness[i] / dist(q, i) for 0 < dist <= radius (float64, the Python loop's operations in the Python loop's order);
 * neighbours[q] = how many keys took part (0: the reference's `dens` is still the int 0 it started as).  One distance launch
 * into pinned memory, the radius filter and the sums on the host in C -- a ten-query environment step against 3000 seen
 * sequences spent 37 us in NumPy on the 30 000-byte matrix.  fitness: one float64 per stored key. */
int fx_cache_density(fx_cache *c, int mode, const uint8_t *queries, int64_t Q, int radius, const double *fitness,
                     double *density, int32_t *neighbours);

/* ------------------------------------------------------- table landscapes */
/* Ground-truth look-up landscapes, e.g. TFBinding
 * (flexs/landscapes/tf_binding.py:38-44: dict of all 4^8 8-mers -> normalised E-score).
 * table[idx] with idx = sum_i code(seq[i]) << (bits*i); entries that do not exist hold NaN
 * (the Python layer turns a NaN / unknown character into the reference's KeyError). */
int fx_table_create(fx_engine *e, const double *table, int64_t len, fx_table **out);
int fx_table_destroy(fx_table *t);
int fx_table_lookup(fx_table *t, const uint8_t *ascii, int64_t N, int L, const uint8_t lut[256],
                    int bits, double *out);
/* Additive landscapes, e.g. AdditiveAAVPackaging._get_raw_fitness
 * (flexs/landscapes/additive_aav_packaging.py:101-107): the table holds L x ncol entries,
 * out[n] = table[0][lut[s0]] + table[1][lut[s1]] + ... accumulated in position order in
 * float64 (the Python `+=` loop's roundings).  lut maps a byte to its column; residues
 * without an entry share an all-zero column.  L <= 3072. */
int fx_table_additive(fx_table *t, const uint8_t *ascii, int64_t N, int L, const uint8_t lut[256],
                      int ncol, double *out);

/* ------------------------------------------------------------ training
 * `KerasModel.train` (flexs/baselines/models/keras_model.py:49-67: model.fit(one_hots, labels, batch_size, epochs) on a
 * model compiled with loss "MSE", optimizer "adam": cnn.py:56, mlp.py:33, global_epistasis_model.py:37) for M ensemble
 * members at once (flexs/ensemble.py:42-52), hand-written for gfx950 (csrc/train_core.h): per mini-batch step one
 * forward+backward launch over (row slices x members) and one Adam launch, enqueued back to back from C.
 * Arithmetic: forward in training mode (Dropout(0.25) before the CNN's last Dense, kept units scaled by 1 / 0.75), MSE
 * over the VALID rows of the mini-batch, reverse-mode gradients (max-pool ties share evenly, ReLU' = 0 at 0), tf.keras
 * Adam (lr 1e-3, beta 0.9 / 0.999, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), w -= lr_t m / (sqrt(v) + 1e-7)).
 * The caller owns weights, moments and the step count (Keras get_weights() order, fx_num_params floats each): they are
 * read at the start and written back at the end, so the optimiser state persists across explorer rounds on the host
 * object as it does on a compiled Keras model (flexs/explorer.py:157-160). */
typedef struct fx_fit_job {
    int kind, L, A, F, H, K;      /* architecture: FX_CNN / FX_MLP / FX_GE (F, K ignored unless CNN) */
    float *weights;               /* in / out */
    float *adam_m, *adam_v;       /* in / out: first / second moments (zeros for a fresh optimiser) */
    int64_t step;                 /* in / out: optimiser step count t */
    const int32_t *order;         /* [epochs][ceil(n / batch)][batch]: data-set row of every mini-batch slot, in the order
                                     the host shuffled them; -1 = padding slot (a partial last mini-batch), valid slots first */
    int epochs, batch;
    const uint8_t *keep;          /* CNN, optional: explicit Dropout keep mask [epochs * steps][batch][H] (1 = keep); NULL =
                                     drawn in the kernel from `seed` (a counter-based hash per (step, slot, unit)) */
    uint64_t seed;
    float *step_loss;             /* out, optional: [epochs * steps] mean squared error of each mini-batch before its update */
} fx_fit_job;
/* ascii: n x L bytes (the measured sequences), labels: n floats.  All jobs train on the same data (same L, same lut).
 * FX_EBADCHAR for a character outside the alphabet (ValueError in the reference's encode loop). */
int fx_train_fit(fx_engine *e, fx_fit_job *jobs, int M, const uint8_t *ascii, int64_t n, int L,
                 const uint8_t lut[256], const float *labels);
/* The epoch shuffles of one fit (`model.fit(..., shuffle=True)`, keras_model.py:62-66 -- Keras draws them from its own
 * generator; any uniform shuffle is the same training procedure): out[e * n .. (e + 1) * n) = a uniformly random permutation of
 * 0 .. n - 1 for every epoch e, all derived from `seed` (Fisher-Yates over xoshiro256** seeded through splitmix64, unbiased
 * bounded draws).  Host only, no engine: 60 shuffles of 1000 rows take 0.12 ms here against 0.5 ms of torch.randperm calls --
 * a seventh of a three-member fit.  Every training path of the Python package (fx_train_fit, the captured-graph and the eager
 * PyTorch steps) takes its orders from here, so a seeded fit shuffles identically on all of them. */
int fx_train_orders(uint64_t seed, int64_t n, int epochs, int32_t *out);

/* ------------------------------------------------------------ test hooks */
/* Host-only (no GPU needed): expose the weight packing (Keras order -> MFMA
 * fragment order) and the bit-parallel Levenshtein that the device kernels use,
 * so the CPU test-suite can check them against the oracle. */
int64_t fx_debug_packed_size(int kind, int L, int A, int F, int H, int K);
int fx_debug_pack_layout(int kind, int L, int A, int F, int H, int K, int64_t *out16);
/* v_mfma_f32_16x16x4_f32 instructions (2048 FLOP each) the MFMA scoring kernels issue per 16-sequence tile per
 * member for this shape -- the kernels' loop bounds restated on the host ('same'-padding taps and the one-hot
 * first layers are not issued, the hidden tail tile runs only its real k-steps).  bench.py prices the ISSUED
 * matrix work of a launch with it, beside the algorithmic FLOP of SURVEY.md 8(d).  Negative: no MFMA kernel. */
int64_t fx_debug_mfma_per_tile(int kind, int L, int A, int F, int H, int K);
/* Profiling aid: `reps` back-to-back fx_score_planes_dev launches bracketed by one hipEvent pair on the engine's
 * stream, issued from C (a Python caller cannot enqueue a < 25 us kernel fast enough to keep the GPU busy, which
 * would be measured as kernel time).  *total_ms = elapsed time of all `reps` launches. */
int fx_debug_time_score(fx_engine *e, fx_model *const *models, int M, const uint8_t *d_ascii, int64_t N, int L,
                        const uint8_t lut[256], float *d_planes, int64_t stride, int reps, float *total_ms);
/* Profiling aid.  With the engine option "train_trace" = 1, fx_train_fit stamps the 100 MHz wall clock after every phase
 * of the forward+backward kernel (workgroup 0 of member 0; the last step's stamps survive): out64[0] start, 1 codes,
 * 2..5 conv1 / conv2 / conv3 / pool (CNN), 20 + i dense layer i forward, 7 loss, 30 + i dense layer i backward, 9..11
 * pool / conv3 / conv2 backward (CNN), 63 end.  0 = not reached. */
int fx_debug_train_trace(fx_engine *e, uint64_t *out64);
/* Profiling aid: `reps` back-to-back neighbour searches (key reset + K4 min-distance kernel) of Q host queries against
 * the device-resident cache, one hipEvent pair on the engine's stream; queries are uploaded once, outside the bracket.
 * *total_ms = elapsed time of all `reps` launches.  1 <= Q <= 32768, cache not empty. */
int fx_debug_time_min_dist(fx_cache *c, int mode, const uint8_t *queries, int64_t Q, int reps, float *total_ms);
/* Profiling aid.  With the engine option "trace" = 1 the MFMA scoring kernels stamp an in-kernel timeline of the LAST
 * launch: row (workgroup b, wave w) = 16 uint64 words at [(b * 16 + w) * 16]: 0 kernel entry, 1 weights resident in
 * LDS, 2 first tile begins, 3 first tile done, 4 last tile done, 5 tiles processed by the wave, 6 wave exit, 7 SIMD
 * id + 1, 8..10 phases of the first tile (first layer / conv part done, next layer done, ...); times are ticks of the
 * 100 MHz constant clock, 0 = never reached.  Copies min(cap_words, 1024*16*16) words. */
int fx_debug_trace_read(fx_engine *e, uint64_t *out, int64_t cap_words);
int fx_debug_pack_weights(int kind, int L, int A, int F, int H, int K, const float *blob, int64_t n,
                          float *packed, int64_t cap);
int fx_debug_myers(const uint8_t *a, int la, const uint8_t *b, int lb);
/* Host only: the banded bounded-distance routine of fx_cache_density's kernel (csrc/myers.h fx_bounded_distance) --
 * min(levenshtein(a, b), K + 1) for K = 1 .. 3 (hamming != 0: position-wise mismatches over rows NUL-padded to a common width). */
int fx_debug_bounded_distance(const uint8_t *a, int la, const uint8_t *b, int lb, int K, int hamming);
/* The strip form of the same recurrence (patterns beyond 768 symbols on the device: csrc/myers.h fx_myers_strip), with
 * strips of 64 x words_per_strip pattern rows (12 = the device's; 1 = test aid, many strip boundaries on short strings). */
int fx_debug_myers_strips(const uint8_t *a, int la, const uint8_t *b, int lb, int words_per_strip);
/* Host only (no GPU): ONE mini-batch training step of one member through the HOST build of csrc/train_core.h -- the
 * source the training kernels are compiled from, threads as loops, the MFMA as an fmaf chain.  The mini-batch is all
 * `rows` rows of `ascii` (<= 4096), cut into slices of R rows as the kernel would; weights / moments / step updated
 * in place, *loss = mean squared error before the update.  keep: optional [rows][H] dropout mask (CNN). */
int fx_debug_train_step_host(int kind, int L, int A, int F, int H, int K, float *weights, float *adam_m,
                             float *adam_v, int64_t *step, const uint8_t *ascii, int rows, const uint8_t lut[256],
                             const float *labels, const uint8_t *keep, int R, float *loss);
/* GPU: one v_mfma_f32_16x16x4_f32 on per-lane operands a[64], b[64], c[64][4] -> d[64][4]
 * (checks the lane layout the kernels assume on the real hardware). */
int fx_debug_mfma_probe(fx_engine *e, const float *a64, const float *b64, const float *c256, float *d256);

#ifdef __cplusplus
}
#endif
#endif /* FLEXS_AMD_H */
