// Microbenchmark / check: v + v^16 and v + v^32 across the rows of a wave with v_permlane16_swap / v_permlane32_swap
// (gfx950 VALU, no LDS crossbar) against __shfl_xor.  Build: hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xor16_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("" : "+v"(b));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor32_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("" : "+v"(b));
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__global__ void k(float* o) {
  float v = o[threadIdx.x];
  float a = xor16_sum(v);
  float b = xor32_sum(a);
  o[64 + threadIdx.x] = a; o[128 + threadIdx.x] = b;
  o[192 + threadIdx.x] = v + __shfl_xor(v, 16, 64);
  float c = v + __shfl_xor(v, 16, 64);
  o[256 + threadIdx.x] = c + __shfl_xor(c, 32, 64);
}
int main() {
  float h[320]; for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 37) + 0.25f * i;
  float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) { if (h[64 + i] != h[192 + i]) ++bad; if (h[128 + i] != h[256 + i]) ++bad; }
  printf("permlane swap sums: %d mismatches (of 128)\n", bad); return bad != 0;
}
