/*
 * fx_oracle.c -- plain-C CPU restatement of the FLEXS get_fitness hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/libfx_oracle.so
 * and loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg (as the checker / the timed CPU baseline).  The product library
 * (flexs_amd/csrc) never links or calls it.
 *
 * Each function cites the reference file:line (relative to /root/reference) it
 * restates.  Written independently of oracle/ref_np.py (scalar loops instead of
 * NumPy matmuls) so the two oracles cross-check each other.
 *
 * Keras forward: PARITY UNPINNED (TensorFlow is an un-vendored third-party
 * dependency -- setup.py:29 -- that cannot be installed here; semantics restated
 * from Keras' documented layer behaviour).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- distances */

/* Unit-cost Levenshtein == editdistance.eval (third-party C++, setup.py:23,
 * docs/requirements.txt:23 pins 0.5.3); call site noisy_abstract_model.py:51.
 * Two-row dynamic programme. */
int fxo_levenshtein(const uint8_t *a, int la, const uint8_t *b, int lb) {
    if (la == 0) return lb;
    if (lb == 0) return la;
    int *prev = (int *)malloc(sizeof(int) * (size_t)(lb + 1) * 2);
    int *cur = prev + lb + 1;
    for (int j = 0; j <= lb; ++j) prev[j] = j;
    for (int i = 1; i <= la; ++i) {
        cur[0] = i;
        for (int j = 1; j <= lb; ++j) {
            int sub = prev[j - 1] + (a[i - 1] != b[j - 1]);
            int del = prev[j] + 1;
            int ins = cur[j - 1] + 1;
            int m = sub < del ? sub : del;
            cur[j] = m < ins ? m : ins;
        }
        int *t = prev; prev = cur; cur = t;
    }
    int r = prev[lb];
    free(prev < cur ? prev : cur);
    return r;
}

int fxo_hamming(const uint8_t *a, const uint8_t *b, int L) {
    int d = 0;
    for (int i = 0; i < L; ++i) d += a[i] != b[i];
    return d;
}

/* noisy_abstract_model.py:42-60 for a batch of queries against a cache in
 * insertion order: first entry attaining the minimum, early exit at distance 1.
 * mode 0 = Levenshtein (parity default), 1 = Hamming.  C == 0 -> dist 0,
 * argmin -1 (the reference returns the query itself as its neighbour). */
void fxo_min_dist(const uint8_t *q, int64_t Q, const uint8_t *cache, int64_t C, int L, int mode,
                  int32_t *dist, int64_t *argmin) {
    for (int64_t i = 0; i < Q; ++i) {
        const uint8_t *s = q + i * L;
        if (C == 0) { dist[i] = 0; argmin[i] = -1; continue; }
        int best = INT32_MAX; int64_t arg = -1;
        for (int64_t c = 0; c < C; ++c) {
            const uint8_t *t = cache + c * L;
            int d = mode == 0 ? fxo_levenshtein(s, L, t, L) : fxo_hamming(s, t, L);
            if (d == 1) { best = 1; arg = c; break; }
            if (d < best) { best = d; arg = c; }
        }
        dist[i] = best; argmin[i] = arg;
    }
}

/* ------------------------------------------------------------ Keras forward */

static inline double relu(double x) { return x > 0.0 ? x : 0.0; }

/* Conv1D(strides=1), cross-correlation, kernel [k][Cin][Cout] (cnn.py:25-47).
 * same: pad_left=(k-1)/2, remaining zeros on the right.  in/out [L][C]. */
static void conv1d(const double *in, int Lin, int Cin, const float *w, const float *b, int k,
                   int Cout, int same, double *out, int Lout) {
    int pl = same ? (k - 1) / 2 : 0;
    for (int t = 0; t < Lout; ++t)
        for (int o = 0; o < Cout; ++o) {
            double acc = b[o];
            for (int j = 0; j < k; ++j) {
                int p = t + j - pl;
                if (p < 0 || p >= Lin) continue;
                for (int c = 0; c < Cin; ++c)
                    acc += in[p * Cin + c] * (double)w[((size_t)j * Cin + c) * Cout + o];
            }
            out[t * Cout + o] = relu(acc);
        }
}

static void dense(const double *in, int nin, const float *w, const float *b, int nout, int act,
                  double *out) {
    for (int o = 0; o < nout; ++o) {
        double acc = b[o];
        for (int i = 0; i < nin; ++i) acc += in[i] * (double)w[(size_t)i * nout + o];
        out[o] = act ? relu(acc) : acc;
    }
}

/* cnn.py:23-54 at predict time.  codes (N, L) alphabet indices; blob = Keras
 * get_weights() order, flattened.  Returns -1 if L < K (valid conv). */
int fxo_cnn_forward(const uint8_t *codes, int64_t N, int L, int A, int F, int H, int K,
                    const float *blob, double *out) {
    if (L < K) return -1;
    int K3 = A - 1, L1 = L - K + 1;
    const float *w1 = blob, *b1 = w1 + (size_t)K * A * F;
    const float *w2 = b1 + F, *b2 = w2 + (size_t)K * F * F;
    const float *w3 = b2 + F, *b3 = w3 + (size_t)K3 * F * F;
    const float *d1 = b3 + F, *c1 = d1 + (size_t)F * H;
    const float *d2 = c1 + H, *c2 = d2 + (size_t)H * H;
    const float *d3 = c2 + H, *c3 = d3 + H;
    double *x = (double *)calloc((size_t)L * A, sizeof(double));
    double *h1 = (double *)malloc(sizeof(double) * (size_t)L1 * F * 3);
    double *h2 = h1 + (size_t)L1 * F, *h3 = h2 + (size_t)L1 * F;
    double *p = (double *)malloc(sizeof(double) * ((size_t)F + 2 * (size_t)H + 1));
    double *g1 = p + F, *g2 = g1 + H;
    for (int64_t n = 0; n < N; ++n) {
        memset(x, 0, sizeof(double) * (size_t)L * A);
        for (int l = 0; l < L; ++l) x[l * A + codes[n * L + l]] = 1.0; /* sequence_utils.py:44-47 */
        conv1d(x, L, A, w1, b1, K, F, 0, h1, L1);
        conv1d(h1, L1, F, w2, b2, K, F, 1, h2, L1);
        /* MaxPooling1D(1) = identity (cnn.py:40) */
        conv1d(h2, L1, F, w3, b3, K3, F, 1, h3, L1);
        for (int f = 0; f < F; ++f) {                       /* GlobalMaxPooling1D */
            double m = h3[f];
            for (int t = 1; t < L1; ++t) if (h3[t * F + f] > m) m = h3[t * F + f];
            p[f] = m;
        }
        dense(p, F, d1, c1, H, 1, g1);
        dense(g1, H, d2, c2, H, 1, g2);
        double y;
        dense(g2, H, d3, c3, 1, 0, &y);
        out[n] = y;
    }
    free(x); free(h1); free(p);
    return 0;
}

/* mlp.py:21-31 (first = L*A -> H) and global_epistasis_model.py:26-36
 * (first = L*A -> 1, then 1 -> H).  Flatten index = l*A + a. */
int fxo_mlp_forward(const uint8_t *codes, int64_t N, int L, int A, int H, const float *blob,
                    double *out) {
    const float *d1 = blob, *c1 = d1 + (size_t)L * A * H;
    const float *d2 = c1 + H, *c2 = d2 + (size_t)H * H;
    const float *d3 = c2 + H, *c3 = d3 + (size_t)H * H;
    const float *d4 = c3 + H, *c4 = d4 + H;
    double *g = (double *)malloc(sizeof(double) * 3 * (size_t)H);
    for (int64_t n = 0; n < N; ++n) {
        for (int o = 0; o < H; ++o) {
            double acc = c1[o];
            for (int l = 0; l < L; ++l) acc += (double)d1[((size_t)l * A + codes[n * L + l]) * H + o];
            g[o] = relu(acc);
        }
        dense(g, H, d2, c2, H, 1, g + H);
        dense(g + H, H, d3, c3, H, 1, g + 2 * H);
        double y;
        dense(g + 2 * H, H, d4, c4, 1, 0, &y);
        out[n] = y;
    }
    free(g);
    return 0;
}

int fxo_ge_forward(const uint8_t *codes, int64_t N, int L, int A, int H, const float *blob,
                   double *out) {
    const float *d1 = blob, *c1 = d1 + (size_t)L * A;
    const float *d2 = c1 + 1, *c2 = d2 + H;
    const float *d3 = c2 + H, *c3 = d3 + (size_t)H * H;
    const float *d4 = c3 + H, *c4 = d4 + H;
    double *g = (double *)malloc(sizeof(double) * 2 * (size_t)H);
    for (int64_t n = 0; n < N; ++n) {
        double s = c1[0];
        for (int l = 0; l < L; ++l) s += (double)d1[(size_t)l * A + codes[n * L + l]];
        s = relu(s);
        for (int o = 0; o < H; ++o) g[o] = relu((double)c2[o] + s * (double)d2[o]);
        dense(g, H, d3, c3, H, 1, g + H);
        double y;
        dense(g + H, H, d4, c4, 1, 0, &y);
        out[n] = y;
    }
    free(g);
    return 0;
}

/* --------------------------------------------------------- ensemble reduce */

/* NumPy pairwise summation order for one contiguous float32 row (what
 * np.mean(x, axis=1) in ensemble.py:24 executes per row; checked bit-exact
 * against numpy 2.2 in tests/test_oracle.py). */
static float np_pairwise_f32(const float *a, int64_t n) {
    if (n < 8) {
        float r = 0.f;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
}

void fxo_ensemble_mean_f32(const float *scores_NM, int64_t N, int M, float *out) {
    for (int64_t n = 0; n < N; ++n) out[n] = np_pairwise_f32(scores_NM + n * M, M) / (float)M;
}

/* sequence_utils.py:65-66: per-position np.argmax -- first maximum wins; a NaN
 * counts as the maximum (first NaN wins), as in NumPy. */
void fxo_argmax_decode(const double *one_hot, int64_t P, int L, int A, uint8_t *idx) {
    for (int64_t r = 0; r < P * L; ++r) {
        const double *row = one_hot + r * A;
        int best = 0;
        for (int a = 1; a < A; ++a) if (row[a] > row[best] || (isnan(row[a]) && !isnan(row[best]))) best = a;
        idx[r] = (uint8_t)best;
    }
}
