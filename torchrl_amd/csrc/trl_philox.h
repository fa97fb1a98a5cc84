// Philox4x32-10 + Box-Muller; the same convention as oracle/philox.py:
//   key = (lo32(env_seed), hi32(env_seed)), counter = (c0, c1, block, tag)
//   u = fmaf(float(x >> 8), 2^-24, 2^-25);  z0 = r cos(2 pi u1), z1 = r sin(2 pi u1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TRL_TAG_RESET 0x52535421u
#define TRL_TAG_NOISE 0x4E4F4953u

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float trl_u01(uint32_t x) {
  return fmaf((float)(x >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f);
}

// four N(0,1) for (c0, c1, block, tag) under env_seed
__device__ __forceinline__ void philox_normals4(uint32_t c0, uint32_t c1, uint32_t block, uint32_t tag,
                                                int64_t env_seed, float z[4]) {
  uint32_t x[4];
  philox4x32_10(c0, c1, block, tag, (uint32_t)(env_seed & 0xFFFFFFFFll),
                (uint32_t)((env_seed >> 32) & 0xFFFFFFFFll), x);
#pragma unroll
  for (int p = 0; p < 4; p += 2) {
    const float r = sqrtf(-2.0f * logf(trl_u01(x[p])));
    float s, c;
    sincospif(2.0f * trl_u01(x[p + 1]), &s, &c);
    z[p] = r * c;
    z[p + 1] = r * s;
  }
}
