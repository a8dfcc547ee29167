// Shared helpers for the gfx950 kernels of libugaitnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ugaitnet_hip.h"

#define UGN_LRELU_ALPHA 0.3f  // keras LeakyReLU() default (reference nets/mj_uwyhNets_ba.py:430)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void ugn_set_error(const char* fmt, ...);

#define UGN_CHECK_LAUNCH(name)                                                      \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      ugn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));         \
      return (int)e__;                                                              \
    }                                                                               \
  } while (0)

#define UGN_REQUIRE(cond, ...)                                                      \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      ugn_set_error(__VA_ARGS__);                                                   \
      return UGN_EINVAL;                                                            \
    }                                                                               \
  } while (0)

// (0 < alpha < 1: max(v, alpha * v) IS v > 0 ? v : alpha * v, signed zeros included -- two instructions instead of three)
__device__ __forceinline__ float ugn_lrelu(float v) { return __builtin_fmaxf(v, v * UGN_LRELU_ALPHA); }
__device__ __forceinline__ float ugn_lrelu_slope(float act) { return act > 0.f ? 1.f : UGN_LRELU_ALPHA; }

// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
__device__ __forceinline__ f32x16 ugn_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
