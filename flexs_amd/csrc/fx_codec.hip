// C ABI of libflexs_amd.so, part 3 of 6 (fx_internal.h): the entry points around the scoring calls -- one-hot encode
// (sequence_utils.py:32-47), ensemble reduction (ensemble.py:54-59, adaptive_ensemble.py:97-102), argmax decode
// (sequence_utils.py:50-66) and decode + score as one round trip (cmaes.py:61-67).  Moved out of fx_score.hip in round 6, unchanged.
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "fx_common.h"
#include "fx_internal.h"

extern "C" {

int fx_encode_onehot_dev(fx_engine* e, const uint8_t* d_ascii, int64_t N, int L, const uint8_t lut[256], int A,
                         float* d_one_hot) {
    if (!e || !lut || N < 0 || L < 0 || A < 1) return FX_EINVAL;
    if (N == 0 || L == 0) return FX_OK;
    if (!d_ascii || !d_one_hot) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    int rc = fx_upload_lut(e, lut);
    if (rc) return rc;
    return fx_launch_encode_onehot(e, d_ascii, N, L, A, d_one_hot);
}

int fx_encode_onehot(fx_engine* e, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], int A,
                     float* one_hot) {
    if (!e || !lut || N < 0 || L < 0 || A < 1) return FX_EINVAL;
    if (N == 0 || L == 0) return FX_OK;
    if (!ascii || !one_hot) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const size_t in_bytes = (size_t)N * L, out_bytes = sizeof(float) * (size_t)N * L * A;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, out_bytes, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, ascii, in_bytes, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_encode_onehot(e, (const uint8_t*)d_in, N, L, A, (float*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(one_hot, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

int fx_ensemble_reduce_dev(fx_engine* e, const float* d_scores, int64_t N, int M, const double* weights,
                           float* d_out32, double* d_out64) {
    if (!e || N < 0 || M < 1) return FX_EINVAL;
    if (N == 0) return FX_OK;
    if (!d_scores || (weights ? !d_out64 : !d_out32)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const double* d_w = nullptr;
    if (weights) {
        void* p = nullptr;
        int rc = fx_scratch(e, 3, sizeof(double) * (size_t)M, &p);
        if (rc) return rc;
        FX_HIP(e, hipMemcpyAsync(p, weights, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, e->stream));
        FX_HIP(e, hipStreamSynchronize(e->stream));       // `weights` is caller memory
        d_w = (const double*)p;
    }
    return fx_launch_ensemble_reduce(e, d_scores, N, M, d_w, d_out32, d_out64);
}

int fx_ensemble_reduce(fx_engine* e, const float* scores, int64_t N, int M, const double* weights, float* out32,
                       double* out64) {
    if (!e || N < 0 || M < 1) return FX_EINVAL;
    if (N == 0) return FX_OK;
    if (!scores || (weights ? !out64 : !out32)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const size_t in_bytes = sizeof(float) * (size_t)N * M;
    const size_t out_bytes = (weights ? sizeof(double) : sizeof(float)) * (size_t)N;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if (in_bytes + out_bytes <= (size_t)e->zero_copy_bytes) {
        FxZeroCopy z;
        const size_t o_w = (in_bytes + 15) / 16 * 16;
        if ((rc = fx_zero_copy_buffers(e, o_w + sizeof(double) * (size_t)M, out_bytes, &z))) return rc;
        std::memcpy(z.h_in, scores, in_bytes);
        if (weights) std::memcpy(z.h_in + o_w, weights, sizeof(double) * (size_t)M);
        if ((rc = fx_launch_ensemble_reduce(e, (const float*)z.d_in, N, M, weights ? (const double*)(z.d_in + o_w) : nullptr,
                                            (float*)z.d_out, (double*)z.d_out))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        std::memcpy(weights ? (void*)out64 : (void*)out32, z.h_out, out_bytes);
        return FX_OK;
    }
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, out_bytes, &d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, scores, in_bytes, hipMemcpyHostToDevice, e->stream));
    rc = fx_ensemble_reduce_dev(e, (const float*)d_in, N, M, weights, (float*)d_out, (double*)d_out);
    if (rc) return rc;
    FX_HIP(e, hipMemcpyAsync(weights ? (void*)out64 : (void*)out32, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

int fx_argmax_decode(fx_engine* e, const double* one_hot, int64_t P, int L, int A, const uint8_t* alphabet,
                     uint8_t* out_chars) {
    if (!e || P < 0 || L < 0 || A < 1) return FX_EINVAL;
    const int64_t rows = P * L;
    if (rows == 0) return FX_OK;
    if (!one_hot || !alphabet || !out_chars) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    lp_disarm(e);                                          // (a pre-launched scoring instance holds most CUs until its idle limit: this call's kernels are queued behind it)
    const size_t in_bytes = sizeof(double) * (size_t)rows * A;
    void *d_in = nullptr, *d_out = nullptr, *d_al = nullptr;
    int rc;
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, (size_t)rows, &d_out))) return rc;
    if ((rc = fx_scratch(e, 3, 256, &d_al))) return rc;
    FX_HIP(e, hipMemcpyAsync(d_in, one_hot, in_bytes, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_al, alphabet, (size_t)A, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_launch_argmax_decode(e, (const double*)d_in, rows, A, (const uint8_t*)d_al, (uint8_t*)d_out))) return rc;
    FX_HIP(e, hipMemcpyAsync(out_chars, d_out, (size_t)rows, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return FX_OK;
}

// CMA-ES / DynaPPO population step (cmaes.py:61-67 + 83-93, environments/dyna_ppo.py:144-163): decode P
// solutions to sequences (K6) and score them with the ensemble in ONE device round trip -- the decoded
// characters never leave the GPU between the two steps.
int fx_decode_score(fx_engine* e, fx_model* const* models, int M, const double* one_hot, int64_t P, int L, int A,
                    const uint8_t* alphabet, const uint8_t lut[256], uint8_t* out_chars, float* out_NM,
                    float* out_mean) {
    int rc = validate_models(e, models, M, L, lut);
    if (rc) return rc;
    if (P < 0 || A < 1) return fx_fail(e, FX_EINVAL, "bad population shape");
    if (models[0]->shape.A != A) return fx_fail(e, FX_ESHAPE, "alphabet size does not match the model's");
    if (P == 0 || L == 0) return FX_OK;
    if (!one_hot || !alphabet || !out_chars || (!out_NM && !out_mean)) return fx_fail(e, FX_EINVAL, "null buffer");
    FX_HIP(e, hipSetDevice(e->device));
    const int64_t rows = P * L;
    const size_t in_bytes = sizeof(double) * (size_t)rows * A;
    const size_t nm_bytes = sizeof(float) * (size_t)P * (size_t)M, mean_bytes = sizeof(float) * (size_t)P;
    void *d_in = nullptr, *d_out = nullptr, *d_txt = nullptr;
    if (in_bytes + 256 + nm_bytes + mean_bytes + (size_t)rows <= (size_t)e->zero_copy_bytes) {
        // a CMA-ES / DyNA-PPO population (15-40 members): the three kernels read and write mapped pinned memory, one wait
        FxZeroCopy z;
        const size_t o_chars = (nm_bytes + mean_bytes + 15) / 16 * 16;
        if ((rc = fx_zero_copy_buffers(e, in_bytes + 256, o_chars + (size_t)rows + 16, &z))) return rc;
        std::memcpy(z.h_in, one_hot, in_bytes);
        std::memcpy(z.h_in + in_bytes, alphabet, (size_t)A);
        if ((rc = fx_upload_lut(e, lut))) return rc;
        float* z_NM = (float*)z.d_out;
        float* z_mean = (float*)(z.d_out + nm_bytes);
        uint8_t* z_chars = (uint8_t*)(z.d_out + o_chars);
        if ((rc = fx_launch_argmax_decode(e, (const double*)z.d_in, rows, A, (const uint8_t*)(z.d_in + in_bytes), z_chars))) return rc;
        if ((rc = score_dispatch(e, models, M, z_chars, P, L, z_NM))) return rc;
        if (out_mean && (rc = fx_launch_ensemble_reduce(e, z_NM, P, M, nullptr, z_mean, nullptr))) return rc;
        if ((rc = fx_wait_small(e))) return rc;          // (a completion value polled in pinned memory: wait_for_results)
        if (out_mean) std::memcpy(out_mean, z.h_out + nm_bytes, mean_bytes);
        if (out_NM) std::memcpy(out_NM, z.h_out, nm_bytes);
        std::memcpy(out_chars, z.h_out + o_chars, (size_t)rows);
        return check_deferred(e);
    }
    if ((rc = fx_scratch(e, 0, in_bytes, &d_in))) return rc;
    if ((rc = fx_scratch(e, 1, nm_bytes + mean_bytes, &d_out))) return rc;
    if ((rc = fx_scratch(e, 3, 256 + (size_t)rows + 16, &d_txt))) return rc;
    uint8_t* d_al = (uint8_t*)d_txt;
    uint8_t* d_chars = d_al + 256;
    float* d_NM = (float*)d_out;
    float* d_mean = (float*)((char*)d_out + nm_bytes);
    FX_HIP(e, hipMemcpyAsync(d_in, one_hot, in_bytes, hipMemcpyHostToDevice, e->stream));
    FX_HIP(e, hipMemcpyAsync(d_al, alphabet, (size_t)A, hipMemcpyHostToDevice, e->stream));
    if ((rc = fx_upload_lut(e, lut))) return rc;
    if ((rc = fx_launch_argmax_decode(e, (const double*)d_in, rows, A, d_al, d_chars))) return rc;
    if ((rc = score_dispatch(e, models, M, d_chars, P, L, d_NM))) return rc;
    if (out_mean) {
        if ((rc = fx_launch_ensemble_reduce(e, d_NM, P, M, nullptr, d_mean, nullptr))) return rc;
        FX_HIP(e, hipMemcpyAsync(out_mean, d_mean, mean_bytes, hipMemcpyDeviceToHost, e->stream));
    }
    if (out_NM) FX_HIP(e, hipMemcpyAsync(out_NM, d_NM, nm_bytes, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipMemcpyAsync(out_chars, d_chars, (size_t)rows, hipMemcpyDeviceToHost, e->stream));
    FX_HIP(e, hipStreamSynchronize(e->stream));
    return check_deferred(e);
}

}  // extern "C"
