// Internal to libflexs_amd.so: what the translation units of the C ABI share (round 5: fx_api.hip, 2 200 lines, became
//   fx_engine.hip    library, engine, options, buffers, timers, models
//   fx_score.hip     scoring entry points (fx_score*) and the launch planner
//   fx_codec.hip     one-hot encode, ensemble reduction, argmax decode (+ score)   [round 6]
//   fx_resident.hip  resident small-call server, pre-launched layer-parallel instance, streamed calls, completion waits
//   fx_nam.hip       NoisyAbstractModel: distances, device cache, table landscapes, blend
//   fx_debug.hip     test / profiling hooks
// no behaviour change: the kernels are other translation units and did not move).  Nothing here is part of the ABI; hidden visibility.
#pragma once
#include <cstdint>
#include <cstddef>

#include "fx_common.h"

#define FXI __attribute__((visibility("hidden")))

struct FxZeroCopy { char *h_in, *d_in, *h_out, *d_out; };
// Plane stride of the engine's member-major intermediate for N sequences (floats; 256-byte aligned planes).
static inline int64_t planar_stride_for(int64_t N) { return (N + 63) & ~(int64_t)63; }

extern "C" {
FXI int check_deferred(fx_engine* e);      // fx_engine.hip
FXI int fx_zero_copy_buffers(fx_engine* e, size_t in_bytes, size_t out_bytes, FxZeroCopy* z);      // fx_engine.hip
FXI int fx_wait_small(fx_engine* e);      // fx_resident.hip
FXI void host_mean_planes(const float* pl, int64_t stride, int64_t N, int M, float* out_mean);      // fx_resident.hip
FXI void lp_arm(fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], float* out_dev, int64_t stride, int mode);      // fx_resident.hip
FXI bool lp_armed_matches(const fx_engine* e, fx_model* const* models, int M, int64_t N, int L, const uint8_t lut[256], int64_t stride, int mode, const void* out_dev);      // fx_resident.hip
FXI void lp_disarm(fx_engine* e);      // fx_resident.hip
FXI unsigned* rows_words_ensure(fx_engine* e);      // fx_resident.hip: the line of device memory a launched-first call's packing lanes store into, or null
FXI bool lp_serve_armed(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], float* out_dev, int64_t stride, int mode, void* out_host, size_t out_bytes);      // fx_resident.hip
FXI int server_call(fx_engine* e, fx_model* const* models, int M, const uint8_t* ascii, int64_t N, int L, const uint8_t lut[256], float* out_NM, float* out_mean);      // fx_resident.hip
FXI int64_t server_since(const fx_engine* e);      // fx_resident.hip
FXI void server_stop(fx_engine* e);      // fx_resident.hip
FXI int wait_for_results(fx_engine* e, unsigned want_seq = 0);      // fx_resident.hip
FXI int score_dispatch(fx_engine* e, fx_model* const* models, int M, const uint8_t* d_ascii, int64_t N, int L, float* d_NM, int64_t planar_stride = 0);      // fx_score.hip
FXI int validate_models(fx_engine* e, fx_model* const* models, int M, int L, const uint8_t* lut);      // fx_score.hip
}  // extern "C"
