// K10b -- forward of a D -> 256 -> 256 -> O MLP (D <= 32, O <= 16) in ONE launch, for up to 12 networks at once.
//
// The 256-wide policy / Q networks of SAC, DDPG and TD3 (torchrl/networks/base.py:30-44, nets.py:34-68: two hidden
// nn.Linear + activation, a linear head) ran as three launches of the dense-layer GEMM per pass.  Their hidden
// activations are small enough to stay on chip: a workgroup owns 32 batch rows, keeps the 32 x 256 activation tile of
// the current layer in LDS, and walks W2 in 16-wide k panels through a double-buffered LDS stage, one barrier per panel.
// A panel is only ~0.5 us of MFMAs per SIMD, far less than an L2 round trip under load, so panels are requested 4 - 5
// ahead into registers (first version, one panel ahead: slower than the three separate launches).  8 waves per workgroup
// (wave w owns hidden columns [32 w, 32 w + 32), 128 registers: two workgroups = 4 waves per SIMD) beat 4 waves with two
// accumulators each (measured 19.9 / 48.5 us against 24.8 / 57.2 us for the 2- and 6-network launches of a SAC update;
// the per-layer GEMM launches took 33.7 / 73.9 us).  The head (O <= 16 columns) splits its reduction over the waves and
// folds the partial tiles in wave order.
// h1 / h2 are written to global memory only for the networks whose backward pass needs them (tape): the target networks
// and the next-state policy pass leave nothing behind but their outputs.
// k order and per-element operation order of layers 1 and 2 are those of gemm_f32_kernel (k = 8 q + 4 hi + r ascending,
// bias, activation): bit-identical hidden activations; the head sums four k-quarters, which is a different rounding
// order than the GEMM's single chain.
#include <algorithm>
#include <cstdlib>
#include "trl_common.h"
#include "trl_mlp.h"

#define M3_H 256
#define M3_R 32                      // batch rows per workgroup
#define M3_LDH (M3_H + 4)            // activation tile row stride
#define M3_KP 16                     // W2 panel depth
#define M3_LDP (M3_KP + 4)
#define M3_DIST 5                    // W2 panels requested ahead of their use (16 registers each)
#define M3_LDX 36                    // first-layer reduction padded to 32 (+ 4)
#define M3_OMAX 16
#define M3_MAXG 12
#define M3_PAN (2 * M3_H * M3_LDP)   // floats: the two W2 panels; W1 (256 x 36) and W3 (16 x 260) + the head partials alias them

struct Mlp3Prob {
  const float* x; const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;
  float* h1; float* h2; float* y;    // h1 / h2 nullable
};
struct Mlp3Dev { int M, D, O, act, last_act; Mlp3Prob p[M3_MAXG]; };

__device__ __forceinline__ float m3_act(int act, float v) {
  if (act == TRL_ACT_TANH) return trl_tanh(v);
  if (act == TRL_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

// NW waves per workgroup; wave w owns hidden columns [CW w, CW w + CW), CW = 256 / NW = 32 CB
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void mlp3_fwd_kernel(Mlp3Dev g) {
  constexpr int DIST = NW == 8 ? 4 : M3_DIST;       // panels requested ahead (8 waves hide more themselves)
  constexpr int NT = 64 * NW, CB = M3_H / (32 * NW), CW = 32 * CB, PJ = M3_H * 4 / NT;   // PJ: 16-byte panel slots per thread
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                                   // [M3_R][M3_LDX]
  float* Hs = Xs + M3_R * M3_LDX;                    // [M3_R][M3_LDH]: H1, later H2
  float* Pan = Hs + M3_R * M3_LDH;                   // [2][M3_H][M3_LDP]  |  W1s [M3_H][M3_LDX]  |  W3s [16][M3_LDH] + partials
  const Mlp3Prob& P = g.p[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * M3_R, D = g.D, M = g.M;

  // the first W2 panels are requested before anything else: they travel while X / W1 are staged and layer 1 runs
  const int prow = tid >> 2, ppiece = tid & 3;       // panel slot: rows prow + (NT / 4) j, floats 4 ppiece .. + 3
  f32x4 pw[DIST][PJ];                             // panels p .. p + DIST - 1 in flight (registers)
  const bool w2_vec = (reinterpret_cast<uintptr_t>(P.w2) & 15) == 0;   // (a network at an odd offset of a flat parameter
  auto fetch_panel = [&](int p, f32x4 (&dst)[PJ]) {                     //  block: dword loads)
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const float* src = P.w2 + (size_t)(prow + (NT / 4) * j) * M3_H + M3_KP * p + 4 * ppiece;
      if (w2_vec) dst[j] = *reinterpret_cast<const f32x4*>(src);
      else        dst[j] = f32x4{src[0], src[1], src[2], src[3]};
    }
  };
  auto stash_panel = [&](int buf, const f32x4 (&src)[PJ]) {
    float* dst = Pan + buf * (M3_H * M3_LDP);
#pragma unroll
    for (int j = 0; j < PJ; ++j) *reinterpret_cast<f32x4*>(dst + (prow + (NT / 4) * j) * M3_LDP + 4 * ppiece) = src[j];
  };
  // a panel is 16 MFMAs per SIMD (~0.5 us): one panel of look-ahead does not cover an L2 round trip under load, DIST do
#pragma unroll
  for (int d = 0; d < DIST; ++d) fetch_panel(d, pw[d]);
  // the head's weights (O x 256 <= 4096 floats), requested now, parked in LDS after layer 2
  constexpr int W3R = M3_OMAX * M3_H / NT;
  float w3r[W3R];
#pragma unroll
  for (int u = 0; u < W3R; ++u) {
    const int e = tid + NT * u, o = e >> 8;
    w3r[u] = o < g.O ? P.w3[e] : 0.0f;
  }
  // ---- stage X (32 x D) and W1 (256 x D), zero padded to 32 columns ----
  {
    const int k = tid & 31, rg = tid >> 5;           // NT / 32 row groups
    for (int r = rg; r < M3_R; r += NT / 32)
      Xs[r * M3_LDX + k] = (k < D && row0 + r < M) ? P.x[(size_t)(row0 + r) * D + k] : 0.0f;
    float* W1s = Pan;
    constexpr int RG = NT / 32, U = M3_H / RG >= 8 ? 8 : M3_H / RG;
    for (int r0 = rg; r0 < M3_H; r0 += RG * U) {      // U loads in flight per thread
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = k < D ? P.w1[(size_t)(r0 + RG * u) * D + k] : 0.0f;
#pragma unroll
      for (int u = 0; u < U; ++u) W1s[(r0 + RG * u) * M3_LDX + k] = v[u];
    }
  }
  __syncthreads();

  f32x16 acc[CB];
  auto zero = [&]() {
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  };
  // ---- layer 1: K = D (padded to a multiple of 8) ----
  zero();
  {
    const float* W1s = Pan;
    const int nq = (D + 7) >> 3;
    for (int q = 0; q < nq; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Xs + i * M3_LDX + 8 * q + 4 * hi);
      f32x4 b[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const f32x4*>(W1s + (CW * wave + 32 * c + i) * M3_LDX + 8 * q + 4 * hi);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[c] = mfma32(a[r], b[c][r], acc[c]);
    }
  }
  // bias + activation, lane (i, hi) owns hidden columns CW wave + 32 c + i of rows rowmap(r, hi)
  auto finish_hidden = [&](const float* bias, float* out_global) {
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int n = CW * wave + 32 * c + i;
      const float bb = bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v = m3_act(g.act, acc[c][r] + bb);
        Hs[row * M3_LDH + n] = v;
        if (out_global && row0 + row < M) out_global[(size_t)(row0 + row) * M3_H + n] = v;
      }
    }
  };
  __syncthreads();                                   // W1s is dead
  finish_hidden(P.b1, P.h1);
  stash_panel(0, pw[0]);
  if (DIST < M3_H / M3_KP) fetch_panel(DIST, pw[0]);
  __syncthreads();

  // ---- layer 2: 16 panels of 16 k (fully unrolled: the register ring is indexed statically) ----
  zero();
#pragma unroll
  for (int p = 0; p < M3_H / M3_KP; ++p) {
    const float* Ws = Pan + (p & 1) * (M3_H * M3_LDP);
#pragma unroll
    for (int q = 0; q < M3_KP / 8; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Hs + i * M3_LDH + M3_KP * p + 8 * q + 4 * hi);
      f32x4 b[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const f32x4*>(Ws + (CW * wave + 32 * c + i) * M3_LDP + 8 * q + 4 * hi);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[c] = mfma32(a[r], b[c][r], acc[c]);
    }
    if (p + 1 < M3_H / M3_KP) {
      stash_panel((p + 1) & 1, pw[(p + 1) % DIST]);             // loaded DIST panels ago
      if (p + 1 + DIST < M3_H / M3_KP) fetch_panel(p + 1 + DIST, pw[(p + 1) % DIST]);
    }
    __syncthreads();
  }
  // H2 replaces H1 (every wave is past its last read of H1: the loop ended on a barrier); W3 and the head partials
  // take the panel space
  finish_hidden(P.b2, P.h2);
  float* W3s = Pan;                                  // [16][M3_LDH], rows >= O zero
  float* Red = Pan + M3_OMAX * M3_LDH;               // [NW waves][32 rows][16]
#pragma unroll
  for (int u = 0; u < W3R; ++u) {
    const int e = tid + NT * u;
    W3s[(e >> 8) * M3_LDH + (e & 255)] = w3r[u];
  }
  __syncthreads();

  // ---- head: wave w reduces k in [KW w, KW w + KW), KW = 256 / NW; output columns = lanes i < 16 ----
  constexpr int KW = M3_H / NW;
  f32x16 hacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
#pragma unroll
  for (int q = 0; q < KW / 8; ++q) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(Hs + i * M3_LDH + KW * wave + 8 * q + 4 * hi);
    const f32x4 b = *reinterpret_cast<const f32x4*>(W3s + (i & 15) * M3_LDH + KW * wave + 8 * q + 4 * hi);
#pragma unroll
    for (int r = 0; r < 4; ++r) hacc = mfma32(a[r], b[r], hacc);
  }
  if (i < M3_OMAX) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Red[(wave * M3_R + (r & 3) + 8 * (r >> 2) + 4 * hi) * M3_OMAX + i] = hacc[r];
  }
  __syncthreads();
  for (int e = tid; e < M3_R * M3_OMAX; e += NT) {
    const int row = e >> 4, o = e & 15;
    if (o < g.O && row0 + row < M) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; w += 4)                  // k-quarters (or -eighths) in wave order, four at a time
        v += (Red[((w + 0) * M3_R + row) * M3_OMAX + o] + Red[((w + 1) * M3_R + row) * M3_OMAX + o]) +
             (Red[((w + 2) * M3_R + row) * M3_OMAX + o] + Red[((w + 3) * M3_R + row) * M3_OMAX + o]);
      v += P.b3 ? P.b3[o] : 0.0f;
      P.y[(size_t)(row0 + row) * g.O + o] = m3_act(g.last_act, v);
    }
  }
}

extern "C" int trl_mlp3_forward_ok(int D, int H1, int H2, int O) {
  return D > 0 && D <= 32 && H1 == M3_H && H2 == M3_H && O > 0 && O <= M3_OMAX;
}

extern "C" int trl_mlp3_forward_group_f32(int G, const float* const* x, const float* const* w1, const float* const* b1,
                                          const float* const* w2, const float* const* b2, const float* const* w3,
                                          const float* const* b3, float* const* h1, float* const* h2, float* const* y,
                                          int M, int D, int O, int act, int last_act, void* stream) {
  TRL_REQUIRE(G >= 1 && G <= M3_MAXG, "1..12 networks per launch");
  TRL_REQUIRE(trl_mlp3_forward_ok(D, M3_H, M3_H, O), "shape not covered (D <= 32, hidden 256 x 256, O <= 16)");
  TRL_REQUIRE(M >= 0 && x && w1 && b1 && w2 && b2 && w3 && y, "bad sizes / null pointer array");
  TRL_REQUIRE((act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE) &&
              (last_act == TRL_ACT_TANH || last_act == TRL_ACT_RELU || last_act == TRL_ACT_NONE), "unknown activation");
  if (M == 0) return TRL_OK;
  Mlp3Dev g{};
  g.M = M; g.D = D; g.O = O; g.act = act; g.last_act = last_act;
  for (int k = 0; k < G; ++k) {
    TRL_REQUIRE(x[k] && w1[k] && b1[k] && w2[k] && b2[k] && w3[k] && y[k], "null pointer");
    g.p[k] = Mlp3Prob{x[k], w1[k], b1[k], w2[k], b2[k], w3[k], b3 ? b3[k] : nullptr, h1 ? h1[k] : nullptr,
                      h2 ? h2[k] : nullptr, y[k]};
  }
  constexpr int lds = (M3_R * M3_LDX + M3_R * M3_LDH + M3_PAN) * (int)sizeof(float);
  static_assert(M3_H * M3_LDX <= M3_PAN && M3_OMAX * M3_LDH + 8 * M3_R * M3_OMAX <= M3_PAN, "aliases fit the panel space");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp3_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("mlp3_forward: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  // 8 waves = 8 column blocks of 32 per workgroup (the 4-wave instantiation measured slower: 34.5 / 82 us against 19.9 / 48.5)
  hipLaunchKernelGGL(mlp3_fwd_kernel<8>, dim3(trl_ceil_div(M, M3_R), G), dim3(512), lds, (hipStream_t)stream, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
