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
 *
 * Round 3: big batches are packed by several threads.  A str object is immutable and the list keeps every item alive
 * for the duration of the call, so the workers only READ object headers and character buffers -- no reference counts,
 * no allocation, no Python API call -- and the GIL is released while they run.  One string is one cache miss on a
 * scattered heap object (~1.6 ns per 8-mer single-threaded at N = 1e5, ~10 ns per 90-mer): the loop is latency-bound,
 * which is exactly what several threads hide.  The pool is persistent (mutex + condition variable; the calling thread
 * takes a share itself), workers are created on first use and never joined.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <string.h>
#include <unistd.h>

#define MAX_WORKERS 15
#define PAR_MIN_BYTES (192 << 10)      /* below this one thread is faster than waking the pool */

typedef struct {
    PyObject** items;
    unsigned char* dst;
    Py_ssize_t n, L;
    int status;                        /* first failure in this share (0 = none) */
} Share;

/* Pack items[0 .. n) -> dst; returns 0 / 1 / 2 / 3, or 4 for a legacy (not "ready") str that needs the GIL. */
static int pack_range(PyObject** items, unsigned char* dst, Py_ssize_t n, Py_ssize_t L, int have_gil) {
    for (Py_ssize_t i = 0; i < n; ++i, dst += L) {
        PyObject* s = items[i];
        if (!PyUnicode_Check(s)) return 3;
        if (!PyUnicode_IS_READY(s)) {
            if (!have_gil) return 4;
            if (PyUnicode_READY(s) < 0) { PyErr_Clear(); return 3; }
        }
        if (PyUnicode_GET_LENGTH(s) != L) return 1;
        const int kind = PyUnicode_KIND(s);
        if (kind == PyUnicode_1BYTE_KIND) {
            memcpy(dst, PyUnicode_1BYTE_DATA(s), (size_t)L);
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

/* ---- persistent pool ---- */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_go = PTHREAD_COND_INITIALIZER, g_done = PTHREAD_COND_INITIALIZER;
static Share g_share[MAX_WORKERS];
static unsigned long g_epoch = 0;      /* bumped per job */
static int g_workers = 0;              /* threads created so far */
static int g_active = 0;               /* workers taking part in the current job */
static int g_pending = 0;              /* shares of the current job not finished yet */
static int g_threads_cfg = 0;          /* 0 = auto */
static int g_busy = 0;                 /* a job owns the pool (a second caller -- another Python thread: the GIL is released
                                          while a job runs -- packs its batch on its own thread instead of waiting) */

static void* worker_main(void* arg) {
    const int id = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_mu);
    for (;;) {
        while (g_epoch == seen || id >= g_active) {
            if (g_epoch != seen && id >= g_active) seen = g_epoch;      /* a job this worker has no share in */
            pthread_cond_wait(&g_go, &g_mu);
        }
        seen = g_epoch;
        Share* sh = &g_share[id];
        pthread_mutex_unlock(&g_mu);
        sh->status = pack_range(sh->items, sh->dst, sh->n, sh->L, 0);
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

/* n items -> dst with `threads` threads (the caller is one of them).  Called WITHOUT the GIL. */
static int pack_parallel(PyObject** items, unsigned char* dst, Py_ssize_t n, Py_ssize_t L, int threads) {
    const int helpers = threads - 1;
    pthread_mutex_lock(&g_mu);
    if (g_busy) {
        pthread_mutex_unlock(&g_mu);
        return pack_range(items, dst, n, L, 0);
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
        g_share[w].items = items + at_row;
        g_share[w].dst = dst + at_row * L;
        g_share[w].n = cnt;
        g_share[w].L = L;
        g_share[w].status = 0;
        at_row += cnt;
    }
    g_active = used;
    g_pending = used;
    ++g_epoch;
    if (used) pthread_cond_broadcast(&g_go);
    pthread_mutex_unlock(&g_mu);

    int status = pack_range(items, dst, per < n ? per : n, L, 0);

    pthread_mutex_lock(&g_mu);
    while (g_pending > 0) pthread_cond_wait(&g_done, &g_mu);
    g_active = 0;
    g_busy = 0;
    for (int w = 0; w < used && status == 0; ++w) status = g_share[w].status;   /* first failing share in row order */
    pthread_mutex_unlock(&g_mu);
    return status;
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
    if (threads > 1) {
        Py_BEGIN_ALLOW_THREADS
        status = pack_parallel(items, dst, n, L, threads);
        Py_END_ALLOW_THREADS
    }
    if (threads <= 1 || status == 4)                                    /* small batch, or a legacy str: under the GIL */
        status = pack_range(items, dst, n, L, 1);
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    return PyLong_FromLong(status);
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
    {"set_threads", set_threads, METH_VARARGS, "set_threads(n) -> previous setting; 0 = auto (min(8, cores / 2)), 1 = single-threaded"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_strpack", NULL, -1, methods};

PyMODINIT_FUNC PyInit__strpack(void) {
    pthread_atfork(NULL, NULL, after_fork_child);
    return PyModule_Create(&moduledef);
}
