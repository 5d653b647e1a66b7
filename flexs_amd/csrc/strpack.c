/* CPython helper: list / tuple of N equal-length str  ->  N x L bytes (latin-1 code points).
 *
 * The reference's entry point takes Python strings (flexs/landscape.py:29-45), and at N = 1e5 turning them
 * into the byte matrix the C ABI wants costs more than the GPU work: "".join + encode + the length checks
 * is ~1.1 ms in pure Python.  This walks the sequence once and memcpy's each string's 1-byte buffer.
 *
 *   pack(seqs, L, out[, start, count]) -> 0   rows of seqs[start : start + count] written to `out` (>= count*L bytes)
 *                          1           some item has a length != L                    (caller raises)
 *                          2           some character does not fit one byte           (caller raises)
 *                          3           some item is not a str                         (caller raises)
 *   set_threads(n) -> previous value   worker threads for big batches (0 = auto: min(8, cores / 2))
 *   pack_staged(seqs, L, staging, stages, tile_pitch, lanes, words, base) -> status as pack, 5 = take the plain path
 *                                      round 5: the host half of a call whose kernels are ALREADY running (fx_score_begin_staged):
 *                                      tiles packed stage by stage into a tile-pitched area, progress published through the BAR
 *   lanes_for(bytes) -> threads such a batch gets
 *
 * Round 5: the next object's header (and a short str's characters, which follow it) is prefetched eight items ahead -- a real batch's
 * strings lie scattered over the heap -- and rows of 8 .. 16 characters move as two overlapping 8-byte words instead of a memcpy call.
 *
 * Round 3: big batches are packed by several threads.  A str object is immutable, so the workers only READ object
 * headers and character buffers -- no reference counts, no allocation, no Python API call.  The CALLING thread keeps the
 * GIL for the whole call (round 4, advisor finding): with the GIL released another Python thread could append to /
 * delete from the caller's list while the workers walk its item array (a reallocated ob_item or a freed str = use after
 * free); holding it costs nothing -- the call lasts tens of microseconds and the caller packs a share itself -- and no
 * Python code can run until it returns, so the list and its strings cannot change.  One string is one cache miss on a
 * scattered heap object (~1.6 ns per 8-mer single-threaded at N = 1e5, ~10 ns per 90-mer): the loop is latency-bound,
 * which is exactly what several threads hide.  The pool is persistent (mutex + condition variable; the calling thread
 * takes a share itself), workers are created on first use and never joined.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#define MAX_WORKERS 15
#define PAR_MIN_BYTES (192 << 10)      /* below this one thread is faster than waking the pool */

typedef struct {
    PyObject** items;                  /* pack job: the strings; NULL = an argmax job */
    unsigned char* dst;
    Py_ssize_t n, L;
    int status;                        /* first failure in this share (0 = none) */
    const double* x;                   /* argmax job: n rows of L = A doubles each -> dst[row] = alphabet[argmax] */
    const unsigned char* alphabet;
    /* staged pack job (Q > 0, pack_staged below): every thread sees ALL n items; thread `worker` of `workers` owns the lanes
     * worker, worker + workers, ... of `lanes` */
    int Q, lanes, worker, workers;
    Py_ssize_t pitch;                  /* bytes from one tile's rows to the next tile's */
    volatile unsigned* words;
    unsigned base;
} Share;

/* `one_hot_to_string` (flexs/utils/sequence_utils.py:50-66) for rows of A doubles: dst[r] = alphabet[np.argmax(x[r])].
 * NumPy's rule: the FIRST maximum wins; a NaN is a maximum (the first NaN wins).  Branch-free pass (strict >, so the first
 * maximum stays), rows that hold a NaN are redone with the exact scalar rule. */
static Py_ssize_t argmax_exact(const double* x, Py_ssize_t A) {   /* numpy: if (!(v <= best)) { best = v; idx = i; if (isnan(best)) break; } */
    double best = x[0];
    Py_ssize_t idx = 0;
    if (best == best)
        for (Py_ssize_t i = 1; i < A; ++i)
            if (!(x[i] <= best)) { best = x[i]; idx = i; if (best != best) break; }
    return idx;
}
/* The same with 256-bit vectors (rows of at least 8 doubles: the protein alphabet's 20): the row's maximum with vmaxpd over its
 * four-double pieces (+ the tail), NaNs noted on the way, then the FIRST position that equals the maximum.  vmaxpd of +0 / -0
 * may return either, vcmpeqpd finds both: the first zero wins, as with NumPy's `>`.  A row that holds a NaN is redone with the
 * exact scalar rule.  ~30 instructions per row of 20 instead of ~120. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2")))
static void argmax_rows_avx2(const double* x, Py_ssize_t rows, Py_ssize_t A, const unsigned char* alphabet, unsigned char* dst) {
    const Py_ssize_t A4 = A & ~(Py_ssize_t)3;
    for (Py_ssize_t r = 0; r < rows; ++r, x += A) {
        __m256d vmax = _mm256_loadu_pd(x);
        __m256d nan = _mm256_cmp_pd(vmax, vmax, _CMP_UNORD_Q);
        for (Py_ssize_t k = 4; k < A4; k += 4) {
            const __m256d v = _mm256_loadu_pd(x + k);
            nan = _mm256_or_pd(nan, _mm256_cmp_pd(v, v, _CMP_UNORD_Q));
            vmax = _mm256_max_pd(vmax, v);
        }
        __m128d m2 = _mm_max_pd(_mm256_castpd256_pd128(vmax), _mm256_extractf128_pd(vmax, 1));
        m2 = _mm_max_sd(m2, _mm_unpackhi_pd(m2, m2));
        double m = _mm_cvtsd_f64(m2);
        int has_nan = _mm256_movemask_pd(nan);
        for (Py_ssize_t k = A4; k < A; ++k) { const double v = x[k]; has_nan |= (v != v); m = v > m ? v : m; }
        Py_ssize_t idx = -1;
        if (!has_nan) {
            const __m256d mm = _mm256_set1_pd(m);
            for (Py_ssize_t k = 0; k < A4; k += 4) {
                const int eq = _mm256_movemask_pd(_mm256_cmp_pd(_mm256_loadu_pd(x + k), mm, _CMP_EQ_OQ));
                if (eq) { idx = k + __builtin_ctz((unsigned)eq); break; }
            }
            if (idx < 0) for (Py_ssize_t k = A4; k < A; ++k) if (x[k] == m) { idx = k; break; }
        }
        if (idx < 0) idx = argmax_exact(x, A);
        dst[r] = alphabet[idx];
    }
}
#endif

static void argmax_rows(const double* x, Py_ssize_t rows, Py_ssize_t A, const unsigned char* alphabet, unsigned char* dst) {
    Py_ssize_t r = 0;
#if defined(__x86_64__)
    static int have_avx2 = -1;
    if (have_avx2 < 0) have_avx2 = __builtin_cpu_supports("avx2") ? 1 : 0;
    if (have_avx2 && A >= 8) { argmax_rows_avx2(x, rows, A, alphabet, dst); return; }
#endif
    /* four rows at a time: the (best, index) recurrence of one row is a chain of ~4-cycle selects; four independent chains
     * keep the core busy (A = 20: 250 -> ~60 us for a population of 40 x 237 rows on one thread) */
    for (; r + 4 <= rows; r += 4, x += 4 * A) {
        const double *x0 = x, *x1 = x + A, *x2 = x + 2 * A, *x3 = x + 3 * A;
        double b0 = x0[0], b1 = x1[0], b2 = x2[0], b3 = x3[0];
        Py_ssize_t i0 = 0, i1 = 0, i2 = 0, i3 = 0;
        int nan = (b0 != b0) | (b1 != b1) | (b2 != b2) | (b3 != b3);
        for (Py_ssize_t i = 1; i < A; ++i) {
            const double v0 = x0[i], v1 = x1[i], v2 = x2[i], v3 = x3[i];
            const int g0 = v0 > b0, g1 = v1 > b1, g2 = v2 > b2, g3 = v3 > b3;
            nan |= (v0 != v0) | (v1 != v1) | (v2 != v2) | (v3 != v3);
            i0 = g0 ? i : i0; b0 = g0 ? v0 : b0;
            i1 = g1 ? i : i1; b1 = g1 ? v1 : b1;
            i2 = g2 ? i : i2; b2 = g2 ? v2 : b2;
            i3 = g3 ? i : i3; b3 = g3 ? v3 : b3;
        }
        if (nan) { i0 = argmax_exact(x0, A); i1 = argmax_exact(x1, A); i2 = argmax_exact(x2, A); i3 = argmax_exact(x3, A); }
        dst[r] = alphabet[i0]; dst[r + 1] = alphabet[i1]; dst[r + 2] = alphabet[i2]; dst[r + 3] = alphabet[i3];
    }
    for (; r < rows; ++r, x += A) dst[r] = alphabet[argmax_exact(x, A)];
}

/* Pack items[0 .. n) -> dst; returns 0 / 1 / 2 / 3, or 4 for a legacy (not "ready") str that needs the GIL. */
static int pack_range(PyObject** items, unsigned char* dst, Py_ssize_t n, Py_ssize_t L, int have_gil) {
    /* The str objects of a real batch lie scattered over the heap: object i + 8's header (and, for a short compact str, its characters,
     * which follow the header) is asked for while object i is copied -- the list's pointer array itself is sequential. */
    enum { AHEAD = 8 };
    for (Py_ssize_t i = 0; i < n; ++i, dst += L) {
        PyObject* s = items[i];
        if (i + AHEAD < n) {
            const char* nx = (const char*)items[i + AHEAD];
            __builtin_prefetch(nx, 0, 1);
            __builtin_prefetch(nx + 64, 0, 1);
            if (L > 64) { __builtin_prefetch(nx + 128, 0, 1); if (L > 128) { __builtin_prefetch(nx + 192, 0, 1); __builtin_prefetch(nx + 256, 0, 1); } }
        }
        if (!PyUnicode_Check(s)) return 3;
        if (!PyUnicode_IS_READY(s)) {
            if (!have_gil) return 4;
            if (PyUnicode_READY(s) < 0) { PyErr_Clear(); return 3; }
        }
        if (PyUnicode_GET_LENGTH(s) != L) return 1;
        const int kind = PyUnicode_KIND(s);
        if (kind == PyUnicode_1BYTE_KIND) {
            const unsigned char* src = PyUnicode_1BYTE_DATA(s);
            /* short rows (RNA / DNA landscapes: 8 .. 50 letters): two overlapping fixed-size moves instead of a call into memcpy */
            if (L >= 8 && L <= 16) {
                uint64_t a, b;
                memcpy(&a, src, 8); memcpy(&b, src + L - 8, 8);
                memcpy(dst, &a, 8); memcpy(dst + L - 8, &b, 8);
            } else {
                memcpy(dst, src, (size_t)L);
            }
        } else {
            const void* data = PyUnicode_DATA(s);
            for (Py_ssize_t j = 0; j < L; ++j) {
                const Py_UCS4 c = PyUnicode_READ(kind, data, j);
                if (c > 255) return 2;
                dst[j] = (unsigned char)c;
            }
        }
    }
    return 0;
}

/* Staged packing for a call whose kernels are already running (include/flexs_amd.h fx_score_begin_staged): the staging area is
 * tile-pitched -- the 16 rows of tile t start at t * pitch -- and stage j = the tiles t with t % Q == j, in the order j = 0, 1, ...;
 * lane l of `lanes` packs the l-th of `lanes` equal parts of a stage's tile list and then publishes base + j + 1 in words[l] --
 * device memory behind the write-combining BAR mapping: a store fence in front (the rows are ordinary stores that must be visible
 * first) and one behind (push the word out now).  Whatever happens, every lane ends at base + Q: the kernels must never be left
 * waiting for a call that failed on the host. */
static inline void publish(volatile unsigned* w, unsigned v) {
#if defined(__x86_64__)
    __builtin_ia32_sfence();
    *w = v;
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    *w = v;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
static int run_staged(const Share* sh) {
    const Py_ssize_t N = sh->n, L = sh->L, pitch = sh->pitch, TG = (N + 15) / 16, Q = sh->Q;
    PyObject** items = sh->items;
    int status = 0;
    for (Py_ssize_t j = 0; j < Q && !status; ++j) {
        const Py_ssize_t nj = j < TG ? (TG - j + Q - 1) / Q : 0;          /* tiles j, j + Q, j + 2 Q, ... < TG */
        for (int l = sh->worker; l < sh->lanes && !status; l += sh->workers) {
            const Py_ssize_t i0 = nj * l / sh->lanes, i1 = nj * (l + 1) / sh->lanes;
            for (Py_ssize_t i = i0; i < i1 && !status; ++i) {
                const Py_ssize_t t = j + i * Q, r0 = t * 16;
                const Py_ssize_t cnt = N - r0 < 16 ? N - r0 : 16;
                if (i + 1 < i1) {                                           /* the next tile's objects lie Q tiles further on: ask for the first of them now */
                    const Py_ssize_t nx = r0 + Q * 16;
                    for (Py_ssize_t r = 0; r < 8 && nx + r < N; ++r) __builtin_prefetch((const char*)items[nx + r], 0, 1);
                }
                status = pack_range(items + r0, sh->dst + t * pitch, cnt, L, 0);
            }
        }
        if (!status)
            for (int l = sh->worker; l < sh->lanes; l += sh->workers) publish(sh->words + l, sh->base + (unsigned)j + 1u);
    }
    for (int l = sh->worker; l < sh->lanes; l += sh->workers) publish(sh->words + l, sh->base + (unsigned)Q);
    return status;
}

/* ---- persistent pool ---- */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_go = PTHREAD_COND_INITIALIZER, g_done = PTHREAD_COND_INITIALIZER;
static Share g_share[MAX_WORKERS];
static unsigned long g_epoch = 0;      /* bumped per job */
static int g_workers = 0;              /* threads created so far */
static int g_active = 0;               /* workers taking part in the current job */
static int g_pending = 0;              /* shares of the current job not finished yet */
static int g_threads_cfg = 0;          /* 0 = auto */
static int g_busy = 0;                 /* a job owns the pool (a second caller -- only possible from a sub-interpreter with its own
                                          GIL -- packs its batch on its own thread instead of waiting) */

/* Hot window: a worker that has just finished a share keeps looking for the next job for ~150 us before it sleeps on the condition
 * variable, and the caller looks that long for its helpers' completion -- a CMA-ES loop hands the pool a 1.5 MB argmax every ~75 us,
 * and waking eight sleeping threads (futex, scheduler) was ~7 of the 19 us such a job took.  Machines with fewer than 16 cores keep
 * the sleeping behaviour (a spinning helper could hold the core the caller needs). */
#if defined(__x86_64__)
static inline unsigned long long hot_clock(void) { return __builtin_ia32_rdtsc(); }
#define HOT_TICKS 450000ull            /* ~150 us at 3 GHz */
#define HOT_PAUSE() __builtin_ia32_pause()
#else
static inline unsigned long long hot_clock(void) { return 0; }
#define HOT_TICKS 0ull
#define HOT_PAUSE() ((void)0)
#endif
static int g_hot = -1;                 /* -1 = not decided yet */
static int hot_enabled(void) {
    if (g_hot < 0) g_hot = (HOT_TICKS > 0 && sysconf(_SC_NPROCESSORS_ONLN) >= 16) ? 1 : 0;
    return g_hot;
}

static void* worker_main(void* arg) {
    const int id = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_mu);
    for (;;) {
        while (g_epoch == seen || id >= g_active) {
            if (g_epoch != seen && id >= g_active) seen = g_epoch;      /* a job this worker has no share in */
            if (hot_enabled() && seen != 0) {
                /* look for the next job without the lock for a while (the epoch only ever grows: a stale read just spins on) */
                pthread_mutex_unlock(&g_mu);
                const unsigned long long t0 = hot_clock();
                int found = 0;
                while (hot_clock() - t0 < HOT_TICKS) {
                    if (__atomic_load_n(&g_epoch, __ATOMIC_ACQUIRE) != seen) { found = 1; break; }
                    HOT_PAUSE();
                }
                pthread_mutex_lock(&g_mu);
                if (found) continue;
                if (g_epoch != seen) continue;
            }
            pthread_cond_wait(&g_go, &g_mu);
        }
        seen = g_epoch;
        Share* sh = &g_share[id];
        pthread_mutex_unlock(&g_mu);
        if (sh->Q > 0) sh->status = run_staged(sh);
        else if (sh->items) sh->status = pack_range(sh->items, sh->dst, sh->n, sh->L, 0);
        else { argmax_rows(sh->x, sh->n, sh->L, sh->alphabet, sh->dst); sh->status = 0; }
        pthread_mutex_lock(&g_mu);
        if (--g_pending == 0) pthread_cond_signal(&g_done);
    }
    return NULL;
}

/* A forked child inherits the counters but not the threads: start over with an empty pool there. */
static void after_fork_child(void) {
    pthread_mutex_init(&g_mu, NULL);
    pthread_cond_init(&g_go, NULL);
    pthread_cond_init(&g_done, NULL);
    g_workers = 0; g_active = 0; g_pending = 0; g_busy = 0;
}

static int want_threads(Py_ssize_t bytes) {
    if (bytes < PAR_MIN_BYTES) return 1;
    int t = g_threads_cfg;
    if (t <= 0) {
        long cores = sysconf(_SC_NPROCESSORS_ONLN);
        t = (int)(cores / 2);
        if (t > 8) t = 8;
    }
    if (t > MAX_WORKERS + 1) t = MAX_WORKERS + 1;
    const Py_ssize_t by_size = bytes / (96 << 10);                     /* >= 96 KiB per thread */
    if (t > by_size) t = (int)by_size;
    return t < 1 ? 1 : t;
}

static int run_share(const Share* sh, Py_ssize_t first, Py_ssize_t count) {
    if (sh->Q > 0) return run_staged(sh);                              /* (a staged job's threads all see the whole batch) */
    if (sh->items) return pack_range(sh->items + first, sh->dst + first * sh->L, count, sh->L, 0);
    argmax_rows(sh->x + first * sh->L, count, sh->L, sh->alphabet, sh->dst + first);
    return 0;
}

/* n items -> dst with `threads` threads (the caller, which holds the GIL, is one of them; the helpers never touch the
 * interpreter: have_gil = 0 makes pack_range hand a legacy str back instead of calling into Python). */
static int job_parallel(const Share* job, Py_ssize_t n, int threads);
static int pack_parallel(PyObject** items, unsigned char* dst, Py_ssize_t n, Py_ssize_t L, int threads) {
    Share job;
    memset(&job, 0, sizeof job);
    job.items = items; job.dst = dst; job.L = L;
    return job_parallel(&job, n, threads);
}
static int job_parallel(const Share* job, Py_ssize_t n, int threads) {
    PyObject** items = job->items;
    unsigned char* dst = job->dst;
    const Py_ssize_t L = job->L;
    const int helpers = threads - 1;
    pthread_mutex_lock(&g_mu);
    if (g_busy) {
        pthread_mutex_unlock(&g_mu);
        if (job->Q > 0) { Share all = *job; all.worker = 0; all.workers = 1; return run_staged(&all); }
        return run_share(job, 0, n);
    }
    g_busy = 1;
    while (g_workers < helpers) {                                       /* grow the pool on demand */
        pthread_t th;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        const int rc = pthread_create(&th, &at, worker_main, (void*)(intptr_t)g_workers);
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        ++g_workers;
    }
    const int used = g_workers < helpers ? g_workers : helpers;
    const Py_ssize_t per = (n + used) / (used + 1);
    Py_ssize_t at_row = per < n ? per : n;                              /* the caller takes rows [0, per) */
    for (int w = 0; w < used; ++w) {
        const Py_ssize_t cnt = (at_row + per <= n) ? per : (n - at_row);
        g_share[w] = *job;
        if (job->Q > 0) { g_share[w].worker = w + 1; g_share[w].workers = used + 1; g_share[w].status = 0; continue; }
        if (items) { g_share[w].items = items + at_row; g_share[w].dst = dst + at_row * L; }
        else { g_share[w].x = job->x + at_row * L; g_share[w].dst = dst + at_row; }
        g_share[w].n = cnt;
        g_share[w].status = 0;
        at_row += cnt;
    }
    g_active = used;
    g_pending = used;
    ++g_epoch;
    if (used) pthread_cond_broadcast(&g_go);
    pthread_mutex_unlock(&g_mu);

    int status;
    if (job->Q > 0) { Share mine = *job; mine.worker = 0; mine.workers = used + 1; status = run_staged(&mine); }
    else status = run_share(job, 0, per < n ? per : n);

    if (hot_enabled()) {                                                /* the helpers are about as far as the caller: look before sleeping */
        const unsigned long long t0 = hot_clock();
        while (__atomic_load_n(&g_pending, __ATOMIC_ACQUIRE) > 0 && hot_clock() - t0 < HOT_TICKS) HOT_PAUSE();
    }
    pthread_mutex_lock(&g_mu);
    while (g_pending > 0) pthread_cond_wait(&g_done, &g_mu);
    g_active = 0;
    g_busy = 0;
    for (int w = 0; w < used && status == 0; ++w) status = g_share[w].status;   /* first failing share in row order */
    pthread_mutex_unlock(&g_mu);
    return status;
}

/* pack_staged(seqs, L, staging_address, stages, tile_pitch, lanes, words_address, base) -> status as `pack` (5 = a legacy str object made
 * the helpers give up: the caller abandons the call and takes the plain path).  lanes_for(bytes) -> how many packing threads a batch of that size gets. */
static PyObject* pack_staged(PyObject* self, PyObject* args) {
    PyObject* seqs;
    Py_ssize_t L;
    unsigned long long dst_addr, words_addr;
    int Q, lanes, pitch;
    unsigned int base;
    if (!PyArg_ParseTuple(args, "OnKiiiKI", &seqs, &L, &dst_addr, &Q, &pitch, &lanes, &words_addr, &base)) return NULL;
    PyObject* fast = PySequence_Fast(seqs, "expected a list or tuple of str");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    if (L < 1 || Q < 1 || pitch < 16 * L || lanes < 1 || lanes > 16 || !dst_addr || !words_addr) {
        Py_DECREF(fast);
        PyErr_SetString(PyExc_ValueError, "strpack.pack_staged: bad arguments");
        return NULL;
    }
    Share job;
    memset(&job, 0, sizeof job);
    job.items = PySequence_Fast_ITEMS(fast); job.dst = (unsigned char*)(uintptr_t)dst_addr; job.n = n; job.L = L;
    job.Q = Q; job.pitch = pitch; job.lanes = lanes; job.words = (volatile unsigned*)(uintptr_t)words_addr; job.base = base;
    int threads = want_threads(n * L);
    if (threads > lanes) threads = lanes;
    long status = threads > 1 ? job_parallel(&job, n, threads) : job_parallel(&job, n, 1);
    if (status == 4) status = 5;
    Py_DECREF(fast);
    return PyLong_FromLong(status);
}
static PyObject* lanes_for(PyObject* self, PyObject* args) {
    Py_ssize_t bytes;
    if (!PyArg_ParseTuple(args, "n", &bytes)) return NULL;
    return PyLong_FromLong(want_threads(bytes));
}

static PyObject* pack(PyObject* self, PyObject* args) {
    PyObject* seqs;
    Py_ssize_t L, start = 0, count = -1;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "Onw*|nn", &seqs, &L, &out, &start, &count)) return NULL;
    PyObject* fast = PySequence_Fast(seqs, "expected a list or tuple of str");
    if (!fast) { PyBuffer_Release(&out); return NULL; }
    const Py_ssize_t total = PySequence_Fast_GET_SIZE(fast);
    if (start < 0 || start > total) start = total;
    const Py_ssize_t n = (count < 0 || start + count > total) ? total - start : count;   /* items [start, start + n) */
    PyObject** items = PySequence_Fast_ITEMS(fast) + start;
    long status = 0;
    if (n > 0 && (L < 0 || out.len < n * L)) {
        PyErr_SetString(PyExc_ValueError, "strpack.pack: output buffer too small");
        Py_DECREF(fast); PyBuffer_Release(&out);
        return NULL;
    }
    unsigned char* dst = (unsigned char*)out.buf;
    const int threads = n > 0 ? want_threads(n * L) : 1;
    if (threads > 1)                                                    /* GIL held: nobody can mutate `seqs` meanwhile (see top) */
        status = pack_parallel(items, dst, n, L, threads);
    if (threads <= 1 || status == 4)                                    /* small batch, or a legacy str: under the GIL */
        status = pack_range(items, dst, n, L, 1);
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    return PyLong_FromLong(status);
}

/* ---- explorer-size calls: pack + fx_score in one C call ------------------------------------------------------------
 * A 20-sequence get_fitness is ~11 us inside fx_score (resident form) and used to spend another ~5 us in Python glue
 * around it (NumPy staging array, ctypes argument conversion).  score_small(plan, seqs, out) packs the strings into a
 * stack buffer and calls fx_score through the function pointer the plan carries.  The plan is built once per model list
 * by flexs_amd/_native.py (struct layout below); this module does not link against libflexs_amd.so. */
typedef int (*fx_score_fn)(void* e, void* const* models, int M, const unsigned char* ascii, long long N, int L,
                           const unsigned char* lut, float* out_NM, float* out_mean);
typedef int (*fx_stream_begin_fn)(void* e, void* const* models, int M, long long N, int L, const unsigned char* lut, unsigned char** rows);
typedef int (*fx_stream_rows_fn)(void* e, long long rows);
typedef int (*fx_stream_end_fn)(void* e, int ok, float* out_NM, float* out_mean);
typedef struct {
    void* fn;                  /* fx_score */
    void* engine;
    long long M, L, want;      /* want: 1 = (N, M) matrix, 2 = mean */
    void* models[16];
    unsigned char lut[256];
    /* streamed calls (include/flexs_amd.h fx_score_stream_*): the strings are packed straight into the resident form's mailbox
     * while its first tiles are already being answered.  stream_min: calls of at least that many strings (0 = never). */
    void* stream_begin; void* stream_rows; void* stream_end;
    long long stream_min, stream_step;
} SmallPlan;
#define SMALL_MAX_BYTES 65536          /* = FX_SERVE_BYTES: what one request of the resident form holds */

/* per-thread 64 KiB packing buffer of score_small, freed when its thread ends (round-5 advisor: it leaked per exiting thread) */
static pthread_key_t big_key;
static pthread_once_t big_once = PTHREAD_ONCE_INIT;
static void big_make_key(void) { (void)pthread_key_create(&big_key, free); }
static unsigned char* big_buffer(void) {
    (void)pthread_once(&big_once, big_make_key);
    unsigned char* b = (unsigned char*)pthread_getspecific(big_key);
    if (!b) {
        b = (unsigned char*)malloc(SMALL_MAX_BYTES);
        if (b && pthread_setspecific(big_key, b) != 0) { free(b); b = NULL; }
    }
    return b;
}

static PyObject* score_small(PyObject* self, PyObject* args) {
    Py_buffer plan, out;
    PyObject* seqs;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*Ow*", &plan, &seqs, &out)) return NULL;
    long status = -1;                                      /* -1: not for this path (the caller takes the general one) */
    if (plan.len == (Py_ssize_t)sizeof(SmallPlan) && (PyList_Check(seqs) || PyTuple_Check(seqs))) {
        const SmallPlan* p = (const SmallPlan*)plan.buf;
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(seqs);
        const Py_ssize_t need = (p->want == 1 ? n * p->M : n) * (Py_ssize_t)sizeof(float);
        if (n > 0 && n * p->L <= SMALL_MAX_BYTES && out.len >= need && p->M >= 1 && p->M <= 16) {
            /* explorer-size calls pack into 4 KiB on the stack; the few that are larger (up to one mailbox request, 64 KiB) into a
             * per-thread heap buffer -- a 64 KiB frame in every call overflowed threads with small stacks (round-4 advisor) */
            unsigned char small[4096];
            unsigned char* buf = small;
            if (n * p->L > (Py_ssize_t)sizeof(small)) buf = big_buffer();
            int changed = 0;                               /* the caller's list was resized while the GIL was released (see below) */
            float* o = (float*)out.buf;
            int streamed = 0;
            if (p->stream_begin && p->stream_min > 0 && n >= p->stream_min && p->stream_step > 0) {
                unsigned char* rows = NULL;
                int rc_begin;
                Py_BEGIN_ALLOW_THREADS                     /* (it may end / start a generation: a stream synchronise; round-4 advisor) */
                rc_begin = ((fx_stream_begin_fn)p->stream_begin)(p->engine, (void* const*)p->models, (int)p->M, (long long)n, (int)p->L, p->lut, &rows);
                Py_END_ALLOW_THREADS
                /* another Python thread ran meanwhile: a list it resized has a stale n (and perhaps moved items) -- give the request
                 * back and leave this call to the general path, which takes its own snapshot (round-5 advisor) */
                changed = PySequence_Fast_GET_SIZE(seqs) != n;
                if (rc_begin == 0 && changed) {
                    ((fx_stream_end_fn)p->stream_end)(p->engine, 0, NULL, NULL);
                    streamed = 1;                          /* (status stays -1) */
                } else if (rc_begin == 0) {
                    /* the request is posted: pack in pieces, front to back, each piece reported as soon as it is in place */
                    PyObject** items = PySequence_Fast_ITEMS(seqs);
                    int st = 0;
                    for (Py_ssize_t r0 = 0; r0 < n && !st; r0 += (Py_ssize_t)p->stream_step) {
                        const Py_ssize_t cnt = n - r0 < (Py_ssize_t)p->stream_step ? n - r0 : (Py_ssize_t)p->stream_step;
                        st = pack_range(items + r0, rows + r0 * p->L, cnt, (Py_ssize_t)p->L, 1);
                        if (!st && r0 + cnt < n) ((fx_stream_rows_fn)p->stream_rows)(p->engine, (long long)(r0 + cnt));
                    }
                    if (st) {
                        ((fx_stream_end_fn)p->stream_end)(p->engine, 0, NULL, NULL);
                        status = 1000 + st;
                        streamed = 1;
                    } else {
                        int rc;
                        Py_BEGIN_ALLOW_THREADS
                        rc = ((fx_stream_end_fn)p->stream_end)(p->engine, 1, p->want == 1 ? o : NULL, p->want == 1 ? NULL : o);
                        Py_END_ALLOW_THREADS
                        if (rc != -7 /* FX_EUNSUPPORTED: the generation went away -- pack into own memory and launch, below */) {
                            status = rc < 0 ? 2000 - rc : rc;
                            streamed = 1;
                        }
                    }
                }
            }
            if (!streamed && PySequence_Fast_GET_SIZE(seqs) != n) { changed = 1; streamed = 1; }   /* (stream_end released the GIL too) */
            const int st = (streamed || !buf) ? 0 : pack_range(PySequence_Fast_ITEMS(seqs), buf, n, (Py_ssize_t)p->L, 1);
            if (streamed) {
                /* answered (or failed) above; `changed`: status is still -1, the general path */
            } else if (!buf) {
                status = -1;                               /* (no memory for the packing buffer: the general path) */
            } else if (st) {
                status = 1000 + st;                        /* 1001 ragged, 1002 non-latin-1, 1003 not a str */
            } else {
                int rc;
                Py_BEGIN_ALLOW_THREADS
                rc = ((fx_score_fn)p->fn)(p->engine, (void* const*)p->models, (int)p->M, buf, (long long)n, (int)p->L, p->lut,
                                          p->want == 1 ? o : NULL, p->want == 1 ? NULL : o);
                Py_END_ALLOW_THREADS
                status = rc;                               /* 0 or an FX_E* code (negative codes are mapped by the caller) */
                if (rc < 0) status = 2000 - rc;
            }
        }
    }
    PyBuffer_Release(&plan);
    PyBuffer_Release(&out);
    return PyLong_FromLong(status);
}

/* decode_argmax(x, rows, A, alphabet, out) -> 0: out[r] = alphabet[np.argmax(x[r * A : (r + 1) * A])] for a C-contiguous float64
 * buffer -- the explorers' `_soln_to_string` / `one_hot_to_string` for a whole population on the HOST.  A population of 40
 * 237-residue solutions is 1.5 MB of doubles: moving them to the GPU for the argmax kernel costs ~100 us of staging copies,
 * several times the scoring launch; here the rows are split over the packing threads (~10 us), and only rows x 1 byte go on. */
static PyObject* decode_argmax(PyObject* self, PyObject* args) {
    Py_buffer x, alpha, out;
    Py_ssize_t rows, A;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*nny*w*", &x, &rows, &A, &alpha, &out)) return NULL;
    long status = 0;
    if (rows < 0 || A < 1 || x.len < rows * A * (Py_ssize_t)sizeof(double) || alpha.len < A || out.len < rows) status = 1;
    else if (rows > 0) {
        Share job;
        memset(&job, 0, sizeof job);
        job.x = (const double*)x.buf; job.alphabet = (const unsigned char*)alpha.buf; job.dst = (unsigned char*)out.buf; job.L = A;
        int threads = want_threads(rows * A * (Py_ssize_t)sizeof(double) / 4);      /* (~4 cycles per double: a thread per 24 KiB-equivalents) */
        if (threads > 1) status = job_parallel(&job, rows, threads);
        else argmax_rows(job.x, rows, A, job.alphabet, job.dst);
    }
    PyBuffer_Release(&x); PyBuffer_Release(&alpha); PyBuffer_Release(&out);
    return PyLong_FromLong(status);
}

/* population_step(plan, x, rows, A, alphabet, chars, scores) -> (status, list of str): the explorers' decode-then-score inner step
 * (cmaes.py:61-67, 83-93; environments/dyna_ppo.py:144-163) in ONE C call: per-position argmax of the (P, L, A) float64 array
 * into `chars` (P x L bytes, as decode_argmax), fx_score of those rows through the plan (as score_small), and the rows as Python
 * strings.  status 0 ok / 1 bad arguments / an FX error code (2000 + |code| if negative).  ~10 us of Python glue (ctypes
 * argument conversion, per-row bytes -> str) of a 100 us CMA-ES step. */
static PyObject* population_step(PyObject* self, PyObject* args) {
    Py_buffer plan, x, alpha, chars, out;
    Py_ssize_t rows, A;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*y*nny*w*w*", &plan, &x, &rows, &A, &alpha, &chars, &out)) return NULL;
    long status = 1;
    PyObject* list = NULL;
    const SmallPlan* p = (const SmallPlan*)plan.buf;
    if (plan.len == (Py_ssize_t)sizeof(SmallPlan) && rows > 0 && A >= 1 && p->L >= 1 && p->M >= 1 && p->M <= 16 &&
        x.len >= rows * p->L * A * (Py_ssize_t)sizeof(double) && alpha.len >= A && chars.len >= rows * p->L &&
        out.len >= (p->want == 1 ? rows * p->M : rows) * (Py_ssize_t)sizeof(float)) {
        const Py_ssize_t L = (Py_ssize_t)p->L, cells = rows * L;
        Share job;
        memset(&job, 0, sizeof job);
        job.x = (const double*)x.buf; job.alphabet = (const unsigned char*)alpha.buf; job.dst = (unsigned char*)chars.buf; job.L = A;
        const int threads = want_threads(cells * A * (Py_ssize_t)sizeof(double) / 4);
        status = 0;
        if (threads > 1) status = job_parallel(&job, cells, threads);
        else argmax_rows(job.x, cells, A, job.alphabet, job.dst);
        if (status == 0) {
            int rc;
            float* o = (float*)out.buf;
            Py_BEGIN_ALLOW_THREADS
            rc = ((fx_score_fn)p->fn)(p->engine, (void* const*)p->models, (int)p->M, (const unsigned char*)chars.buf, (long long)rows, (int)L, p->lut,
                                      p->want == 1 ? o : NULL, p->want == 1 ? NULL : o);
            Py_END_ALLOW_THREADS
            status = rc < 0 ? 2000 - rc : rc;
        }
        if (status == 0) {
            list = PyList_New(rows);
            for (Py_ssize_t r = 0; list && r < rows; ++r) {
                PyObject* str = PyUnicode_DecodeLatin1((const char*)chars.buf + r * L, L, NULL);
                if (!str) { Py_CLEAR(list); break; }
                PyList_SET_ITEM(list, r, str);
            }
            if (!list) { PyBuffer_Release(&plan); PyBuffer_Release(&x); PyBuffer_Release(&alpha); PyBuffer_Release(&chars); PyBuffer_Release(&out); return NULL; }
        }
    }
    PyBuffer_Release(&plan); PyBuffer_Release(&x); PyBuffer_Release(&alpha); PyBuffer_Release(&chars); PyBuffer_Release(&out);
    if (!list) { list = Py_None; Py_INCREF(list); }
    return Py_BuildValue("(lN)", status, list);
}

/* adalead_children(nodes, mu, alphabet, seen_before, seen_now, random, getrandbits) -> (child_idxs, children) | None
 * One tree level of Adalead's roll-outs (flexs/baselines/explorers/adalead.py:128-150, flexs_amd/utils/rollouts.py _children): a
 * child per node -- the parent of child number k is nodes[k - 1], so the first child descends from the LAST node -- drawn again
 * until it is in neither `seen_before` (a set) nor `seen_now` (a dict).  A child is generate_random_mutant(node, mu, alphabet)
 * (sequence_utils.py:87-108): per residue one random(); below mu the residue becomes random.choice(alphabet), i.e.
 * alphabet[_randbelow(len(alphabet))] with _randbelow(n) = getrandbits(n.bit_length()) redrawn until it is below n.
 * EVERY draw goes through the two callables the caller passes (the `random` module's own bound methods): the module's stream is
 * consumed exactly as the Python loop consumes it -- the explorer traces generated by the reference pin that.  The loop itself
 * (41 % of a round: ~1.2 us per mutant in Python) is what moves to C.  None: an argument this path does not handle (the caller
 * runs the Python loop). */
static PyObject* adalead_children(PyObject* self, PyObject* args) {
    PyObject *nodes, *alphabet, *seen_before, *seen_now, *rnd, *bits;
    double mu;
    (void)self;
    if (!PyArg_ParseTuple(args, "OdOOOOO", &nodes, &mu, &alphabet, &seen_before, &seen_now, &rnd, &bits)) return NULL;
    if (!PyList_CheckExact(nodes) || !PyUnicode_Check(alphabet) || !PyAnySet_CheckExact(seen_before) || !PyDict_CheckExact(seen_now) ||
        PyUnicode_READY(alphabet) < 0 || PyUnicode_KIND(alphabet) != PyUnicode_1BYTE_KIND)
        Py_RETURN_NONE;
    const Py_ssize_t n_nodes = PyList_GET_SIZE(nodes), n_alpha = PyUnicode_GET_LENGTH(alphabet);
    if (n_alpha < 1 || n_alpha > 255) Py_RETURN_NONE;
    const unsigned char* alpha = PyUnicode_1BYTE_DATA(alphabet);
    int k_bits = 0;
    for (Py_ssize_t t = n_alpha; t; t >>= 1) ++k_bits;                 /* n.bit_length() */
    for (Py_ssize_t i = 0; i < n_nodes; ++i) {
        PyObject* nd = PyList_GET_ITEM(nodes, i);
        if (!PyTuple_CheckExact(nd) || PyTuple_GET_SIZE(nd) != 2 || !PyUnicode_Check(PyTuple_GET_ITEM(nd, 1)) ||
            PyUnicode_READY(PyTuple_GET_ITEM(nd, 1)) < 0 || PyUnicode_KIND(PyTuple_GET_ITEM(nd, 1)) != PyUnicode_1BYTE_KIND ||
            PyUnicode_GET_LENGTH(PyTuple_GET_ITEM(nd, 1)) > 4096 || PyUnicode_GET_LENGTH(PyTuple_GET_ITEM(nd, 1)) < 1)
            Py_RETURN_NONE;                                             /* (an empty node: the Python loop raises what the reference raises) */
    }
    PyObject* k_obj = PyLong_FromLong(k_bits);
    PyObject* idxs = PyList_New(0);
    PyObject* children = PyList_New(0);
    if (!k_obj || !idxs || !children) goto fail;
    unsigned char buf[4096];
    while (PyList_GET_SIZE(children) < n_nodes) {
        const Py_ssize_t have = PyList_GET_SIZE(children);
        PyObject* nd = PyList_GET_ITEM(nodes, have == 0 ? n_nodes - 1 : have - 1);      /* nodes[len(children) - 1] */
        PyObject* node = PyTuple_GET_ITEM(nd, 1);
        const Py_ssize_t L = PyUnicode_GET_LENGTH(node);
        const unsigned char* src = PyUnicode_1BYTE_DATA(node);
        const double mu_eff = mu * 1 / (double)L;                        /* mu * 1 / len(node), evaluated as Python does */
        for (Py_ssize_t j = 0; j < L; ++j) {
            PyObject* r = PyObject_CallNoArgs(rnd);
            if (!r) goto fail;
            const double x = PyFloat_AsDouble(r);
            Py_DECREF(r);
            if (x == -1.0 && PyErr_Occurred()) goto fail;
            if (x < mu_eff) {
                long pick;
                do {
                    PyObject* b = PyObject_CallOneArg(bits, k_obj);
                    if (!b) goto fail;
                    pick = PyLong_AsLong(b);
                    Py_DECREF(b);
                    if (pick == -1 && PyErr_Occurred()) goto fail;
                } while (pick >= n_alpha);
                buf[j] = alpha[pick];
            } else {
                buf[j] = src[j];
            }
        }
        PyObject* child = PyUnicode_DecodeLatin1((const char*)buf, L, NULL);
        if (!child) goto fail;
        int seen = PySet_Contains(seen_before, child);
        if (seen == 0) seen = PyDict_Contains(seen_now, child);
        if (seen < 0) { Py_DECREF(child); goto fail; }
        if (seen == 0) {
            if (PyList_Append(idxs, PyTuple_GET_ITEM(nd, 0)) < 0 || PyList_Append(children, child) < 0) { Py_DECREF(child); goto fail; }
        }
        Py_DECREF(child);
    }
    Py_DECREF(k_obj);
    return Py_BuildValue("(NN)", idxs, children);
fail:
    Py_XDECREF(k_obj); Py_XDECREF(idxs); Py_XDECREF(children);
    return NULL;
}

static PyObject* set_threads(PyObject* self, PyObject* args) {
    int n;
    if (!PyArg_ParseTuple(args, "i", &n)) return NULL;
    pthread_mutex_lock(&g_mu);
    const int prev = g_threads_cfg;
    g_threads_cfg = n < 0 ? 0 : n;
    pthread_mutex_unlock(&g_mu);
    return PyLong_FromLong(prev);
}

static PyMethodDef methods[] = {
    {"pack", pack, METH_VARARGS, "pack(seqs, L, out[, start, count]) -> status (0 ok, 1 ragged, 2 non-latin-1 character, 3 not a str)"},
    {"pack_staged", pack_staged, METH_VARARGS, "pack_staged(seqs, L, staging_address, stages, tile_pitch, lanes, words_address, base) -> status (as pack; 5 = legacy str objects: abandon, take the plain path)"},
    {"lanes_for", lanes_for, METH_VARARGS, "lanes_for(bytes) -> packing threads a batch of that many bytes gets"},
    {"score_small", score_small, METH_VARARGS, "score_small(plan, seqs, out) -> 0 ok, -1 not applicable, 1001..1003 packing status, FX error code (2000 + |code| if negative)"},
    {"decode_argmax", decode_argmax, METH_VARARGS, "decode_argmax(x_float64, rows, A, alphabet_bytes, out_uint8) -> 0 ok, 1 bad arguments: out[r] = alphabet[argmax of row r] (NumPy's first-max / NaN rule)"},
    {"population_step", population_step, METH_VARARGS, "population_step(plan, x_float64, rows, A, alphabet_bytes, chars_uint8, scores_float32) -> (status, list of str | None): argmax decode + fx_score through the plan + the rows as str"},
    {"adalead_children", adalead_children, METH_VARARGS, "adalead_children(nodes, mu, alphabet, seen_before_set, seen_now_dict, random.random, random.getrandbits) -> (child_idxs, children) | None: one tree level of Adalead's roll-outs, every draw through the two callables"},
    {"set_threads", set_threads, METH_VARARGS, "set_threads(n) -> previous setting; 0 = auto (min(8, cores / 2)), 1 = single-threaded"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_strpack", NULL, -1, methods};

PyMODINIT_FUNC PyInit__strpack(void) {
    pthread_atfork(NULL, NULL, after_fork_child);
    return PyModule_Create(&moduledef);
}
