// Shared by simt_ref.cpp (the host build of csrc/train_core.h) and simt_train.cpp (its device branches under the SIMT emulator): one
// seeded training problem -- network shape, weights, a mini-batch with some padding slots, dropout masks, labels.
#pragma once
#include <cstdint>
#include <vector>

struct SimtProblem {
    int kind, L, A, F, H, K, rows, R;
    std::vector<float> w, labels;
    std::vector<int32_t> order;
    std::vector<uint8_t> ascii, lut, keep;
};

inline SimtProblem simt_problem(int kind, int L, int A, int F, int H, int K, int rows, int R, int P, unsigned seed) {
    unsigned long long s = seed * 2654435761ull + 12345;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); };
    SimtProblem p{kind, L, A, F, H, K, rows, R, {}, {}, {}, {}, {}, {}};
    p.w.resize((size_t)P);
    for (auto& x : p.w) x = ((int)(rnd() % 2001) - 1000) * 1e-4f;
    p.order.resize((size_t)rows);
    for (int i = 0; i < rows; ++i) p.order[(size_t)i] = (rnd() % 7 == 0) ? -1 : (int)(rnd() % rows);
    p.ascii.resize((size_t)rows * L);
    p.lut.assign(256, 0xFF);
    for (int a = 0; a < A; ++a) p.lut[65 + a] = (uint8_t)a;
    for (auto& c : p.ascii) c = (uint8_t)(65 + rnd() % A);
    p.keep.resize((size_t)rows * H);
    for (auto& k : p.keep) k = rnd() % 4 != 0;
    p.labels.resize((size_t)rows);
    for (auto& y : p.labels) y = (rnd() % 1000) * 1e-3f;
    return p;
}

// One step through the HOST build (simt_ref.cpp): the updated weights, and the gradient partials of every slice.
std::vector<float> simt_ref_step(const SimtProblem& p, std::vector<float>* partials);
int simt_ref_params(int kind, int L, int A, int F, int H, int K);
