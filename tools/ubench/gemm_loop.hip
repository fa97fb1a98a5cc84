// Microbenchmark: the inner loop of gemm_f32_kernel (csrc/k_gemm.hip) with the operand panels resident in LDS -- cycles per
// v_mfma_f32_32x32x2_f32 for (a) one 32x32 accumulator per wave (one A and one B ds_read_b128 per 4 MFMAs: the kernel's
// loop), (b) a 64x32 register block (two accumulators: 2 A + 1 B reads per 8 MFMAs), (c) a 64x64 block (4 accumulators:
// 2 A + 2 B reads per 16 MFMAs), at 1 and 2 workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 gemm_loop.hip -o gemm_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KC 128
#define LDK (KC + 4)
template <int RA, int RB>                      // register blocks along the A rows / B rows of a wave
__global__ __launch_bounds__(256) void k(float* out, int panels, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int GA = 64 * RA, GB = 64 * RB;     // 2 x 2 waves
  float* As = lds;
  float* Bs = lds + GA * LDK;
  for (int e = threadIdx.x; e < (GA + GB) * LDK; e += 256) lds[e] = seed + (e & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  f32x16 acc[RA][RB];
  for (int a = 0; a < RA; ++a) for (int b = 0; b < RB; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (int p = 0; p < panels; ++p) {
#pragma unroll 1
    for (int q4 = 0; q4 < KC / 32; ++q4) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int q = 4 * q4 + qq;
        f32x4 av[RA], bv[RB];
#pragma unroll
        for (int a = 0; a < RA; ++a) av[a] = *reinterpret_cast<const f32x4*>(As + (32 * (wm * RA + a) + i) * LDK + 8 * q + 4 * hi);
#pragma unroll
        for (int b = 0; b < RB; ++b) bv[b] = *reinterpret_cast<const f32x4*>(Bs + (32 * (wn * RB + b) + i) * LDK + 8 * q + 4 * hi);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][r], bv[b][r], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();                            // (the kernel has a barrier per panel)
  }
  float s = 0;
  for (int a = 0; a < RA; ++a) for (int b = 0; b < RB; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int RA, int RB> void run(const char* name, int wgs, float* d, int panels) {
  const int lds = (64 * RA + 64 * RB) * LDK * 4;
  hipFuncSetAttribute((const void*)k<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RA, RB>), dim3(wgs), dim3(256), lds, 0, d, panels, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RA, RB>), dim3(wgs), dim3(256), lds, 0, d, panels, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)panels * (KC / 2) * RA * RB * (wgs / 256);     // per wave x waves per SIMD
  const double flops = (double)wgs * 4 * panels * (KC / 2) * RA * RB * 4096.0;
  printf("%-34s %4d WGs (LDS %3d KB): %8.1f us  %6.1f ns per MFMA per SIMD (27.9 ideal at 2.3 GHz)  %6.1f TFLOP/s\n", name, wgs, lds >> 10,
         ms * 1e3, ms * 1e6 / mfma_per_simd, flops / (ms * 1e-3) * 1e-12);
}
int main() {
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  const int panels = 400;
  run<1, 1>("32x32 per wave (kernel today)", 256, d, panels);
  run<1, 1>("32x32 per wave (kernel today)", 512, d, panels);
  run<2, 1>("64x32 per wave", 256, d, panels);
  run<2, 1>("64x32 per wave", 512, d, panels / 2);
  run<2, 2>("64x64 per wave", 256, d, panels / 2);
  return 0;
}
