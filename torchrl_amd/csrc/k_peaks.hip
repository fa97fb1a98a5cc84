// Calibration kernels for the two rooflines the path is priced against ("measure achievable with a copy
// kernel on the box and quote both" / "quote measured peak").  Not part of the hot path: bench.py runs them once after
// its timed region and reports `peaks_measured` next to the nominal 8 TB/s / 157.3 TFLOP/s.
//   copy   16-byte loads and stores on every CU (three launch shapes, the caller quotes the best): bytes moved = 2 * n * 4
//   mfma   register-resident v_mfma_f32_32x32x2_f32 issue loop, four independent accumulators per wave, no memory
//          traffic inside the loop: the ceiling a kernel made of nothing but fp32 matrix instructions reaches
#include "trl_common.h"

// mode 0: one pass, a workgroup copies ONE contiguous 16 KB piece (4 x 16 bytes per lane, all loads before the first store),
//         as many workgroups as pieces -- the hardware dispatcher balances the CUs
// mode 1: the same with non-temporal loads and stores
// mode 2: persistent grid (16 workgroups per CU), grid-stride, 4 loads in flight per lane, non-temporal
template <bool NT>
__global__ __launch_bounds__(256) void peak_copy_piece_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                              int64_t n_vec) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + 256 * k;
    if (i < n_vec) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + 256 * k;
    if (i < n_vec) { if (NT) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
  }
}

__global__ __launch_bounds__(256) void peak_copy_stride_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                               int64_t n_vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const f32x4 a = __builtin_nontemporal_load(src + i);
    const f32x4 b = __builtin_nontemporal_load(src + i + stride);
    const f32x4 c = __builtin_nontemporal_load(src + i + 2 * stride);
    const f32x4 d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride);
    __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n_vec; i += stride) dst[i] = src[i];
}

// mode 3: 4-byte accesses (one dword per lane and instruction, 4 in flight) -- not a bandwidth figure: the pattern of the
// gradient kernel's input loads, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for that access width
__global__ __launch_bounds__(256) void peak_copy_dword_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int64_t i = base + 256 * k; v[k] = i < n ? src[i] : 0.0f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int64_t i = base + 256 * k; if (i < n) dst[i] = v[k]; }
}

extern "C" int trl_peak_copy_f32(const float* src, float* dst, int64_t n, int mode, void* stream) {
  TRL_REQUIRE(src && dst && n > 0 && (n & 3) == 0, "src / dst non-null, n a positive multiple of 4");
  TRL_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "16-byte aligned pointers");
  TRL_REQUIRE(mode >= 0 && mode <= 3, "mode 0..3");
  const int64_t n_vec = n / 4;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 3) {
    const int64_t wg = (n + 1023) / 1024;
    TRL_REQUIRE(wg < (1ll << 31), "too large");
    hipLaunchKernelGGL(peak_copy_dword_kernel, dim3((unsigned)wg), dim3(256), 0, s, src, dst, n);
  } else if (mode == 2) {
    int64_t wg = (n_vec + 255) / 256;
    if (wg > 256 * 16) wg = 256 * 16;
    hipLaunchKernelGGL(peak_copy_stride_kernel, dim3((unsigned)wg), dim3(256), 0, s, (const f32x4*)src, (f32x4*)dst, n_vec);
  } else {
    const int64_t wg = (n_vec + 1023) / 1024;
    TRL_REQUIRE(wg < (1ll << 31), "too large");
    if (mode == 0) hipLaunchKernelGGL(peak_copy_piece_kernel<false>, dim3((unsigned)wg), dim3(256), 0, s, (const f32x4*)src, (f32x4*)dst, n_vec);
    else           hipLaunchKernelGGL(peak_copy_piece_kernel<true>, dim3((unsigned)wg), dim3(256), 0, s, (const f32x4*)src, (f32x4*)dst, n_vec);
  }
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

__global__ __launch_bounds__(256) void peak_mfma_kernel(float* __restrict__ out, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = 1.0f + 1e-3f * (float)(threadIdx.x & 63), b = 1.0f - 1e-3f * (float)(threadIdx.x & 31);
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;     // keeps the accumulators alive
}

// FLOPs of one call = workgroups * 4 waves * iters * 4 MFMAs * (2 * 32 * 32 * 2); out: workgroups * 256 floats
extern "C" int trl_peak_mfma_f32(float* out, int workgroups, int iters, void* stream) {
  TRL_REQUIRE(out && workgroups > 0 && iters > 0, "out non-null, positive sizes");
  hipLaunchKernelGGL(peak_mfma_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, out, iters);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
