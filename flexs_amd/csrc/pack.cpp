// Host-side weight packing: Keras get_weights() order -> MFMA fragment layout.
// Pure C++ (no device calls) so it is unit-testable on a machine without a GPU
// through fx_debug_pack_weights().  Layout documented in fx_common.h / DESIGN.md.
#include <cstring>
#include <vector>

#include "fx_common.h"

static inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

int64_t fx_num_params(const FxShape& s) {
    const int64_t L = s.L, A = s.A, F = s.F, H = s.H, K = s.K;
    switch (s.kind) {
        case FX_CNN:   // cnn.py:23-54
            return K * A * F + F + K * F * F + F + (A - 1) * F * F + F + F * H + H + H * H + H + H + 1;
        case FX_MLP:   // mlp.py:21-31
            return L * A * H + H + H * H + H + H * H + H + H + 1;
        case FX_GE:    // global_epistasis_model.py:26-36
            return L * A + 1 + H + H + H * H + H + H + 1;
    }
    return -1;
}

int fx_hidden_pos(int h, int H) {
    const int HT = (H + 15) / 16, base = 16 * (HT - 1);
    if (h < base) return h;
    const int i = h - base;
    return base + 4 * (i % 4) + i / 4;
}

FxPackLayout fx_pack_layout(const FxShape& s) {
    FxPackLayout p{};
    p.FT = (s.F + 15) / 16;
    p.HTR = (s.H + 15) / 16;
    p.HT = p.HTR;
    {
        // the MFMA kernels are instantiated for these tile counts; round up (extra tiles are zero padding
        // that the kernels skip); beyond the largest one the layout stays exact and the generic kernels run
        static const int sizes[] = {1, 2, 4, 7, 8, 13, 16};
        for (int v : sizes) if (v >= p.HTR) { p.HT = v; break; }
    }
    {
        const int tail = s.H - 16 * (p.HTR - 1);
        p.RLH = tail >= 4 ? (tail + 3) / 4 : 1;          // unit i sits in k-step i / 4
        if (p.RLH > 4) p.RLH = 4;
    }
    const int64_t BLK = 256;
    int64_t off = 0;
    if (s.kind == FX_CNN) {
        p.SG1 = (s.K * s.A + 15) / 16;
        p.off_first = off; off += (int64_t)p.SG1 * p.FT * BLK;
        p.off_c2 = off;    off += (int64_t)s.K * p.FT * p.FT * BLK;
        p.off_c3 = off;    off += (int64_t)s.K3() * p.FT * p.FT * BLK;
        p.off_cb = off;    off += 3 * 16 * p.FT;
        p.off_w1p = off;   off += (int64_t)s.K * s.A * FX_C1_ROW(p.FT);
        p.conv_floats = off = rup(off, 4);
        p.off_d1 = off;    off += (int64_t)p.FT * p.HT * BLK;
        p.off_d2 = off;    off += (int64_t)p.HT * p.HT * BLK;
        p.off_d3 = off;
        p.off_db = off;    off += 3 * 16 * p.HT + 4;          // bd1, bd2, w3, bout
    } else if (s.kind == FX_MLP) {
        p.SG1 = (s.L * s.A + 15) / 16;
        p.off_first = off; off += (int64_t)p.SG1 * p.HT * BLK;
        p.off_c2 = p.off_c3 = p.off_cb = off;
        p.off_w1p = off;   off += (int64_t)s.L * s.A * 16 * p.HT;   // layer-1 kernel, plain [L*A][16HT] rows (gather form)
        p.conv_floats = off;
        p.off_d1 = off;
        p.off_d2 = off;    off += (int64_t)p.HT * p.HT * BLK;
        p.off_d3 = off;    off += (int64_t)p.HT * p.HT * BLK;
        p.off_db = off;    off += 4 * 16 * p.HT + 4;          // b1, b2, b3, w4, bout
    } else {                                                   // FX_GE
        p.SG1 = 0;
        p.off_first = off; off += rup((int64_t)s.L * s.A, 4); // w1 vector, plain
        p.off_c2 = p.off_c3 = p.off_cb = off;
        p.conv_floats = off;
        p.off_d1 = p.off_d2 = off;
        p.off_d3 = off;    off += (int64_t)p.HT * p.HT * BLK;
        p.off_db = off;    off += 4 + 4 * 16 * p.HT + 4;      // b1[4], w2, b2, b3, w4, bout
    }
    p.total_floats = rup(off, 4);
    p.off_w1pair = -1;
    p.pair_floats = 0;
    if (s.kind == FX_MLP && s.A == 4) {
        p.off_w1pair = p.total_floats;
        // rows are FX_PAIR_PAD floats apart more than their length: 16 lanes gather 16 different rows at once, and with a
        // stride of 16 HT floats (112: 48 mod 64 banks) rows r and r + 4 start on the same LDS bank
        p.pair_floats = ((int64_t)(s.L / 2) * 16 + (s.L % 2) * 4) * (16 * p.HT + FX_PAIR_PAD);
    }
    p.alloc_floats = p.total_floats + p.pair_floats;
    return p;
}

// v_mfma_f32_16x16x4_f32 instructions the MFMA scoring kernels issue per 16-sequence tile per member (the kernels'
// own loop bounds restated on the host: conv taps that fall into the 'same' zero padding are skipped, the one-hot
// first layers run as gathers, the hidden tail tile runs RLH of its 4 k-steps).  Used by bench.py / tests to price
// the ISSUED matrix work of a launch; -1 when no MFMA kernel covers the shape.
int64_t fx_mfma_per_tile(const FxShape& s) {
    const FxPackLayout p = fx_pack_layout(s);
    if (p.HTR > 16) return -1;
    // one HxH layer: every output tile runs the k-steps of all input tiles, the last real one only RLH of them
    // (a hidden size that was rounded up to a larger instantiated tile count runs all padded k-steps)
    const int64_t hh = (int64_t)p.HT * (p.HT == p.HTR ? 4 * (p.HT - 1) + p.RLH : 4 * p.HT);
    if (s.kind == FX_MLP) return 2 * hh;
    if (s.kind == FX_GE) return hh;
    if (s.kind != FX_CNN || s.L < s.K) return -1;
    const int L1 = s.L1(), K3 = s.K3();
    const int PL2 = (s.K - 1) / 2, PL3 = (K3 - 1) / 2;
    int64_t taps2 = 0, taps3 = 0;
    for (int t = 0; t < L1; ++t) {
        for (int j = 0; j < s.K; ++j) taps2 += (t + j - PL2 >= 0 && t + j - PL2 < L1);
        for (int j = 0; j < K3; ++j) taps3 += (t + j - PL3 >= 0 && t + j - PL3 < L1);
    }
    const int64_t per_tap = (int64_t)p.FT * p.FT * 4;
    return (taps2 + taps3) * per_tap + (int64_t)p.FT * 4 * p.HT + hh;
}

namespace {
// Maps from a POSITION in the padded layout to the source index (or -1 = zero padding).
struct PosMap {
    std::vector<int> src;
    static PosMap identity(int n, int padded) {
        PosMap m; m.src.assign(padded, -1);
        for (int i = 0; i < n; ++i) m.src[i] = i;
        return m;
    }
    static PosMap hidden(int H) {
        const int HT = (H + 15) / 16;
        PosMap m; m.src.assign(16 * HT, -1);
        for (int h = 0; h < H; ++h) m.src[fx_hidden_pos(h, H)] = h;
        return m;
    }
    int operator()(int pos) const { return pos < (int)src.size() ? src[pos] : -1; }
};

// W is [rows][cols] row-major.  dense-style block: input position = 16*mi + 4*g + r.
void pack_dense_block(const float* W, int cols, const PosMap& rmap, const PosMap& cmap, int mi, int mo, float* blk) {
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int kin = rmap(16 * mi + 4 * (lane >> 4) + r);
            const int o = cmap(16 * mo + (lane & 15));
            blk[lane * 4 + r] = (kin >= 0 && o >= 0) ? W[(int64_t)kin * cols + o] : 0.f;
        }
}
// first-layer block: input row = 16*sg + 4*r + g.
void pack_first_block(const float* W, int rows, int cols, const PosMap& cmap, int sg, int mo, float* blk) {
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int kin = 16 * sg + 4 * r + (lane >> 4);
            const int o = cmap(16 * mo + (lane & 15));
            blk[lane * 4 + r] = (kin < rows && o >= 0) ? W[(int64_t)kin * cols + o] : 0.f;
        }
}
void pack_vec(const float* v, const PosMap& map, int padded, float* dst) {
    for (int i = 0; i < padded; ++i) dst[i] = map(i) >= 0 ? v[map(i)] : 0.f;
}
}  // namespace

void fx_pack_weights(const FxShape& s, const float* blob, float* packed) {
    const FxPackLayout p = fx_pack_layout(s);
    std::memset(packed, 0, sizeof(float) * (size_t)p.alloc_floats);
    const int A = s.A, F = s.F, H = s.H, K = s.K, L = s.L, FT = p.FT, HT = p.HT;
    const PosMap hid = PosMap::hidden(H), fil = PosMap::identity(F, 16 * FT);
    const float* w = blob;
    if (s.kind == FX_CNN) {
        const int K3 = s.K3();
        const float* w1 = w;  w += (int64_t)K * A * F;
        const float* b1 = w;  w += F;
        const float* w2 = w;  w += (int64_t)K * F * F;
        const float* b2 = w;  w += F;
        const float* w3 = w;  w += (int64_t)K3 * F * F;
        const float* b3 = w;  w += F;
        const float* d1 = w;  w += (int64_t)F * H;
        const float* c1 = w;  w += H;
        const float* d2 = w;  w += (int64_t)H * H;
        const float* c2 = w;  w += H;
        const float* d3 = w;  w += H;
        const float* c3 = w;
        for (int sg = 0; sg < p.SG1; ++sg)
            for (int mo = 0; mo < FT; ++mo)
                pack_first_block(w1, K * A, F, fil, sg, mo, packed + p.off_first + ((int64_t)sg * FT + mo) * 256);
        for (int j = 0; j < K; ++j)
            for (int mi = 0; mi < FT; ++mi)
                for (int mo = 0; mo < FT; ++mo)
                    pack_dense_block(w2 + (int64_t)j * F * F, F, fil, fil, mi, mo,
                                     packed + p.off_c2 + (((int64_t)j * FT + mi) * FT + mo) * 256);
        for (int j = 0; j < K3; ++j)
            for (int mi = 0; mi < FT; ++mi)
                for (int mo = 0; mo < FT; ++mo)
                    pack_dense_block(w3 + (int64_t)j * F * F, F, fil, fil, mi, mo,
                                     packed + p.off_c3 + (((int64_t)j * FT + mi) * FT + mo) * 256);
        pack_vec(b1, fil, 16 * FT, packed + p.off_cb);
        pack_vec(b2, fil, 16 * FT, packed + p.off_cb + 16 * FT);
        pack_vec(b3, fil, 16 * FT, packed + p.off_cb + 32 * FT);
        for (int k = 0; k < K * A; ++k) pack_vec(w1 + (int64_t)k * F, fil, 16 * FT, packed + p.off_w1p + (int64_t)k * FX_C1_ROW(FT));
        for (int mi = 0; mi < FT; ++mi)
            for (int mo = 0; mo < HT; ++mo)
                pack_dense_block(d1, H, fil, hid, mi, mo, packed + p.off_d1 + ((int64_t)mi * HT + mo) * 256);
        for (int mi = 0; mi < HT; ++mi)
            for (int mo = 0; mo < HT; ++mo)
                pack_dense_block(d2, H, hid, hid, mi, mo, packed + p.off_d2 + ((int64_t)mi * HT + mo) * 256);
        pack_vec(c1, hid, 16 * HT, packed + p.off_db);
        pack_vec(c2, hid, 16 * HT, packed + p.off_db + 16 * HT);
        pack_vec(d3, hid, 16 * HT, packed + p.off_db + 32 * HT);
        packed[p.off_db + 48 * HT] = c3[0];
    } else if (s.kind == FX_MLP) {
        const float* d1 = w;  w += (int64_t)L * A * H;
        const float* c1 = w;  w += H;
        const float* d2 = w;  w += (int64_t)H * H;
        const float* c2 = w;  w += H;
        const float* d3 = w;  w += (int64_t)H * H;
        const float* c3 = w;  w += H;
        const float* d4 = w;  w += H;
        const float* c4 = w;
        for (int sg = 0; sg < p.SG1; ++sg)
            for (int mo = 0; mo < HT; ++mo)
                pack_first_block(d1, L * A, H, hid, sg, mo, packed + p.off_first + ((int64_t)sg * HT + mo) * 256);
        for (int mi = 0; mi < HT; ++mi)
            for (int mo = 0; mo < HT; ++mo) {
                pack_dense_block(d2, H, hid, hid, mi, mo, packed + p.off_d2 + ((int64_t)mi * HT + mo) * 256);
                pack_dense_block(d3, H, hid, hid, mi, mo, packed + p.off_d3 + ((int64_t)mi * HT + mo) * 256);
            }
        for (int k = 0; k < L * A; ++k) pack_vec(d1 + (int64_t)k * H, hid, 16 * HT, packed + p.off_w1p + (int64_t)k * 16 * HT);
        pack_vec(c1, hid, 16 * HT, packed + p.off_db);
        pack_vec(c2, hid, 16 * HT, packed + p.off_db + 16 * HT);
        pack_vec(c3, hid, 16 * HT, packed + p.off_db + 32 * HT);
        pack_vec(d4, hid, 16 * HT, packed + p.off_db + 48 * HT);
        packed[p.off_db + 64 * HT] = c4[0];
        if (p.off_w1pair >= 0) {
            // one row per pair of positions and pair of letters: the float32 sum of the two single-position rows
            const int64_t R = 16 * HT, RP = R + FX_PAIR_PAD;        // (the pad floats stay 0)
            const float* rows = packed + p.off_w1p;
            float* dst = packed + p.off_w1pair;
            for (int pi = 0; pi < L / 2; ++pi)
                for (int c0 = 0; c0 < 4; ++c0)
                    for (int c1 = 0; c1 < 4; ++c1)
                        for (int64_t k = 0; k < R; ++k)
                            dst[((int64_t)pi * 16 + 4 * c0 + c1) * RP + k] =
                                rows[((int64_t)(2 * pi) * 4 + c0) * R + k] + rows[((int64_t)(2 * pi + 1) * 4 + c1) * R + k];
            if (L % 2)
                for (int c0 = 0; c0 < 4; ++c0)
                    for (int64_t k = 0; k < R; ++k)
                        dst[((int64_t)(L / 2) * 16 + c0) * RP + k] = rows[((int64_t)(L - 1) * 4 + c0) * R + k];
        }
    } else {
        const float* d1 = w;  w += (int64_t)L * A;
        const float* c1 = w;  w += 1;
        const float* d2 = w;  w += H;
        const float* c2 = w;  w += H;
        const float* d3 = w;  w += (int64_t)H * H;
        const float* c3 = w;  w += H;
        const float* d4 = w;  w += H;
        const float* c4 = w;
        pack_vec(d1, PosMap::identity(L * A, (int)rup((int64_t)L * A, 4)), (int)rup((int64_t)L * A, 4), packed + p.off_first);
        for (int mi = 0; mi < HT; ++mi)
            for (int mo = 0; mo < HT; ++mo)
                pack_dense_block(d3, H, hid, hid, mi, mo, packed + p.off_d3 + ((int64_t)mi * HT + mo) * 256);
        float* db = packed + p.off_db;
        db[0] = c1[0];
        pack_vec(d2, hid, 16 * HT, db + 4);
        pack_vec(c2, hid, 16 * HT, db + 4 + 16 * HT);
        pack_vec(c3, hid, 16 * HT, db + 4 + 32 * HT);
        pack_vec(d4, hid, 16 * HT, db + 4 + 48 * HT);
        db[4 + 64 * HT] = c4[0];
    }
}
