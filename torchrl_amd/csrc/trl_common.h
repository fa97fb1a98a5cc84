// Shared host/device helpers for libtrl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/trl_hip.h"

void trl_set_error(const char* fmt, ...);

#define TRL_REQUIRE(cond, msg)                                        \
  do { if (!(cond)) { trl_set_error("%s: %s", __func__, msg); return TRL_EINVAL; } } while (0)

#define TRL_LAUNCH_CHECK()                                            \
  do { hipError_t e_ = hipGetLastError();                             \
       if (e_ != hipSuccess) { trl_set_error("%s: %s", __func__, hipGetErrorString(e_)); \
                               return (int)e_; } } while (0)

static inline int trl_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x2  __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- wave64 reductions (all 64 lanes get the result) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
