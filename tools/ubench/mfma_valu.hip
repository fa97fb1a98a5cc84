// Microbenchmark: v_mfma_f32_16x16x4_f32 throughput per SIMD with 1 / 2 waves per SIMD, alone and beside VALU work
// of the same or the other wave.  Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NM 64
// mode 0: all waves MFMA only; 1: all waves VALU only; 2: every wave MFMA block then VALU block (serial in-order);
// 3: even waves MFMA only, odd waves VALU only (2 waves/SIMD: one of each per SIMD when blockDim = 512)
template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{seed, 0, 0, 0};
  float v[8];
  for (int c = 0; c < 8; ++c) v[c] = seed + c;
  const float a = seed * 0.5f, b = seed * 0.25f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave < 4));
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >= 4));
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int m = 0; m < NM * 6; ++m) v[m & 7] = fmaf(v[m & 7], 1.0001f, 0.5f);   // 6 plain VALU per MFMA slot
    }
  }
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int c = 0; c < 8; ++c) s += v[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int threads, float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf_per_simd = (double)iters * NM * (MODE == 3 ? 1 : threads / 256);   // MFMAs per SIMD
  printf("%-44s threads %3d: %8.1f us   %.1f ns per MFMA-slot per SIMD\n", name, threads, ms * 1e3, ms * 1e6 / mf_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 2000;
  run<0>("MFMA only, 1 wave/SIMD", 256, d, iters);
  run<0>("MFMA only, 2 waves/SIMD", 512, d, iters);
  run<1>("VALU only (6/slot), 1 wave/SIMD", 256, d, iters);
  run<1>("VALU only (6/slot), 2 waves/SIMD", 512, d, iters);
  run<2>("MFMA block then VALU block, 1 wave/SIMD", 256, d, iters);
  run<2>("MFMA block then VALU block, 2 waves/SIMD", 512, d, iters);
  run<3>("wave A MFMA, wave B VALU on each SIMD", 512, d, iters);
  return 0;
}
