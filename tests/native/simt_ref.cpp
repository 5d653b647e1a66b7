// The reference side of the SIMT emulation test: one training step through the HOST build of csrc/train_core.h (threads as loops, the
// MFMA as an fmaf chain) -- the build the CPU suite already holds to oracle/train_np.py.  Kept in a namespace of its own: the other
// translation unit compiles the same header's DEVICE branches.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "simt_common.h"

namespace fxt_ref {
#include "../../flexs_amd/csrc/train_core.h"
}

int simt_ref_params(int kind, int L, int A, int F, int H, int K) { return fxt_ref::fxt_net(kind, L, A, kind == 0 ? F : 0, H, kind == 0 ? K : 0).P; }

std::vector<float> simt_ref_step(const SimtProblem& p, std::vector<float>* partials) {
    using namespace fxt_ref;
    FxtJob j{};
    j.net = fxt_net(p.kind, p.L, p.A, p.kind == 0 ? p.F : 0, p.H, p.kind == 0 ? p.K : 0);
    j.net.ldx = j.net.F;                                   // plain rows: what the rotated / staged forms are measured against
    j.batch = p.rows; j.steps_per_epoch = 1; j.total_steps = 1; j.n = p.rows; j.R = p.R; j.S = (p.rows + p.R - 1) / p.R;
    std::vector<float> w = p.w, m((size_t)j.net.P, 0.f), v((size_t)j.net.P, 0.f), part((size_t)j.S * (j.net.P + 1), 0.f);
    const float lr = 1e-3f;
    float loss = 0.f;
    j.w = w.data(); j.adam_m = m.data(); j.adam_v = v.data(); j.partial = part.data(); j.order = p.order.data();
    j.keep = p.kind == 0 ? p.keep.data() : nullptr; j.lr_t = &lr; j.step_loss = &loss;
    j.ws_slice = fxt_ws(j.net, p.R).total;
    std::vector<float> ws((size_t)j.ws_slice, 0.f);
    for (int s = 0; s < j.S; ++s) {
        std::fill(ws.begin(), ws.end(), 0.f);
        fxt_forward_backward<0, 0>(j, FxtWg{0, 1}, 0, s, p.ascii.data(), p.lut.data(), p.labels.data(), ws.data(), (const float*)j.w);
    }
    if (partials) *partials = part;
    fxt_step_loss(j, 0);
    for (int i = 0; i < j.net.P; ++i) fxt_adam(j, 0, i);
    return w;
}
