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

// Round 4: a workgroup re-reads all of W2 (256 KB, from L2) for its rows, so with 32-row workgroups a 6-network launch
// pulls 196 MB through L2 for 3.5 GFLOP -- as long as its MFMAs take.  Workgroups of 64 rows (RB = 2 accumulator blocks
// per wave: every W2 operand read from LDS feeds two MFMAs, 16 MFMAs per wave and barrier instead of 8) halve that, but
// 64-row tiles alone quantise badly (6 x 64 = 384 workgroups at one per CU = two rounds for 1.5 rounds of work): the
// launch mixes them -- as many networks on 64-row tiles as fill one round of the chip, the rest on 32-row tiles that
// form a second, half-length round.  Same k order per output element: results are bit-identical to the 32-row kernel.
#define M3_H 256
#define M3_R 32                      // batch rows per accumulator block (a workgroup owns RB of them)
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
struct Mlp3Dev { int M, D, O, act, last_act; int n_big, tiles_big, tiles_small; Mlp3Prob p[M3_MAXG]; };   // networks [0, n_big): 64-row tiles

__device__ __forceinline__ float m3_act(int act, float v) {
  if (act == TRL_ACT_TANH) return trl_tanh(v);
  if (act == TRL_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

// NW waves per workgroup; wave w owns hidden columns [CW w, CW w + CW), CW = 256 / NW = 32 CB
// RB: 32-row accumulator blocks per workgroup (rows [row0, row0 + 32 RB) of network `net`)
template <int NW, int RB>
__device__ __forceinline__ void mlp3_body(const Mlp3Dev& g, float* lds, int net, int row0) {
  constexpr int DIST = NW == 8 ? 4 : M3_DIST;       // panels requested ahead (8 waves hide more themselves)
  constexpr int NT = 64 * NW, CB = M3_H / (32 * NW), CW = 32 * CB, PJ = M3_H * 4 / NT;   // PJ: 16-byte panel slots per thread
  constexpr int R = M3_R * RB;                       // rows of this workgroup
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(RB == 1 || CB == 1, "two row blocks per wave go with one column block");
  float* Xs = lds;                                   // [R][M3_LDX]
  float* Hs = Xs + R * M3_LDX;                       // [R][M3_LDH]: H1, later H2
  float* Pan = Hs + R * M3_LDH;                      // [2][M3_H][M3_LDP]  |  W1s [M3_H][M3_LDX]  |  W3s [16][M3_LDH]
  const Mlp3Prob P = g.p[net];                       // by value: a reference is re-read from the argument block inside the
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;   // panel loop (s_load +
  const int D = g.D, M = g.M;                        // lgkmcnt(0): a wait for every pending LDS operation, per panel)

  // the first W2 panels are requested before anything else: they travel while X / W1 are staged and layer 1 runs
  const int prow = tid >> 2, ppiece = tid & 3;       // panel slot: rows prow + (NT / 4) j, floats 4 ppiece .. + 3
  f32x4 pw[DIST][PJ];                             // panels p .. p + DIST - 1 in flight (registers)
  // (a network at an odd offset of a flat parameter block is only 4-byte aligned: the 16-byte load is spelled as a copy
  // from 4-byte-aligned floats, which this target serves with one global_load_dwordx4 either way -- a per-load alignment
  // branch cost four taken branches per panel)
  auto fetch_panel = [&](int p, f32x4 (&dst)[PJ]) {
#pragma unroll
    for (int j = 0; j < PJ; ++j)
      __builtin_memcpy(&dst[j], P.w2 + (size_t)(prow + (NT / 4) * j) * M3_H + M3_KP * p + 4 * ppiece, 16);
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
  // ---- stage X (R x D) and W1 (256 x D), zero padded to 32 columns ----
  // (unconditional loads from clamped addresses + a select: predicated loads cost a branch around each of them)
  {
    const int k = tid & 31, rg = tid >> 5;           // NT / 32 row groups
    const bool kin = k < D;
    const int kc = kin ? k : 0;
    constexpr int XR = R / (NT / 32);                // X rows per thread
    float xv[XR];
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int row = row0 + rg + (NT / 32) * u;
      xv[u] = P.x[(size_t)(row < M ? row : 0) * D + kc];
    }
    float* W1s = Pan;
    constexpr int RG = NT / 32, U = M3_H / RG >= 8 ? 8 : M3_H / RG;
    for (int r0 = rg; r0 < M3_H; r0 += RG * U) {      // U loads in flight per thread
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = P.w1[(size_t)(r0 + RG * u) * D + kc];
#pragma unroll
      for (int u = 0; u < U; ++u) W1s[(r0 + RG * u) * M3_LDX + k] = kin ? v[u] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int r = rg + (NT / 32) * u;
      Xs[r * M3_LDX + k] = (kin && row0 + r < M) ? xv[u] : 0.0f;
    }
  }
  __syncthreads();

  f32x16 acc[RB][CB];
  auto zero = [&]() {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][c][r] = 0.0f;
  };
  // ---- layer 1: K = D (padded to a multiple of 8) ----
  zero();
  {
    const float* W1s = Pan;
    const int nq = (D + 7) >> 3;
    for (int q = 0; q < nq; ++q) {
      f32x4 a[RB], b[CB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(Xs + (M3_R * rb + i) * M3_LDX + 8 * q + 4 * hi);
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const f32x4*>(W1s + (CW * wave + 32 * c + i) * M3_LDX + 8 * q + 4 * hi);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[rb][c] = mfma32(a[rb][r], b[c][r], acc[rb][c]);
    }
  }
  // bias + activation into the LDS tile; lane (i, hi) owns hidden columns CW wave + 32 c + i of rows rowmap(r, hi).
  // The copy a backward pass needs (tape) leaves from the TILE after the next barrier, as whole 16-byte row pieces
  // (4 loads + 4 stores per thread and 32 rows) -- storing the accumulators directly took 16 four-byte stores per lane
  // with their own address arithmetic and row predicates: ~300 vector + ~300 scalar instructions per wave and layer,
  // a third of the kernel's MFMA time on a part whose fp32 MFMAs do not overlap vector work.
  auto finish_hidden = [&](const float* bias) {
    // (the activation switch sits OUTSIDE the element loops: inside, every element carried its own scalar compare and
    // branch chain -- ~5 taken branches per element, a few thousand cycles per wave and layer)
    auto body = [&](auto actf) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int n = CW * wave + 32 * c + i;
        const float bb = bias[n];
        float* dst = Hs + (4 * hi) * M3_LDH + n;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dst[(M3_R * rb + (r & 3) + 8 * (r >> 2)) * M3_LDH] = actf(acc[rb][c][r] + bb);
      }
    };
    if (g.act == TRL_ACT_RELU)      body([](float v) { return fmaxf(v, 0.0f); });
    else if (g.act == TRL_ACT_TANH) body([](float v) { return trl_tanh(v); });
    else                            body([](float v) { return v; });
  };
  const bool rows_full = row0 + R <= M;              // (uniform) no row predicate in the common case
  auto store_tile = [&](float* out_global) {         // Hs (complete: call behind a barrier) -> out_global rows [row0, row0 + R)
    if (!out_global) return;
    constexpr int PT = R * (M3_H / 4) / NT;          // 16-byte pieces per thread
    const int c4 = tid & 63, rbase = tid >> 6;       // piece c4 of rows rbase + (NT / 64) t
    f32x4 v[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) v[t] = *reinterpret_cast<const f32x4*>(Hs + (rbase + (NT / 64) * t) * M3_LDH + 4 * c4);
    float* o = out_global + (size_t)(row0 + rbase) * M3_H + 4 * c4;
    if (rows_full) {
#pragma unroll
      for (int t = 0; t < PT; ++t) *reinterpret_cast<f32x4*>(o + (size_t)(NT / 64) * t * M3_H) = v[t];
    } else {
#pragma unroll
      for (int t = 0; t < PT; ++t)
        if (row0 + rbase + (NT / 64) * t < M) *reinterpret_cast<f32x4*>(o + (size_t)(NT / 64) * t * M3_H) = v[t];
    }
  };
  __syncthreads();                                   // W1s is dead
  finish_hidden(P.b1);
  stash_panel(0, pw[0]);
  if (DIST < M3_H / M3_KP) fetch_panel(DIST, pw[0]);
  __syncthreads();
  store_tile(P.h1);                                  // (its tile reads are long done when H2 overwrites the tile)

  // ---- layer 2: 16 panels of 16 k (fully unrolled: the register ring is indexed statically) ----
  zero();
#pragma unroll
  for (int p = 0; p < M3_H / M3_KP; ++p) {
    const float* Ws = Pan + (p & 1) * (M3_H * M3_LDP);
#pragma unroll
    for (int q = 0; q < M3_KP / 8; ++q) {
      f32x4 a[RB], b[CB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(Hs + (M3_R * rb + i) * M3_LDH + M3_KP * p + 8 * q + 4 * hi);
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const f32x4*>(Ws + (CW * wave + 32 * c + i) * M3_LDP + 8 * q + 4 * hi);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[rb][c] = mfma32(a[rb][r], b[c][r], acc[rb][c]);
    }
    if (p + 1 < M3_H / M3_KP) {
      stash_panel((p + 1) & 1, pw[(p + 1) % DIST]);             // loaded DIST panels ago
      if (p + 1 + DIST < M3_H / M3_KP) fetch_panel(p + 1 + DIST, pw[(p + 1) % DIST]);
    }
    __syncthreads();
  }
  // H2 replaces H1 (every wave is past its last read of H1: the loop ended on a barrier); W3 and the head partials
  // take the panel space
  finish_hidden(P.b2);
  float* W3s = Pan;                                  // [16][M3_LDH], rows >= O zero
  float* Red = Pan + M3_PAN;                         // [NW waves][R rows][16] (behind the panel space)
#pragma unroll
  for (int u = 0; u < W3R; ++u) {
    const int e = tid + NT * u;
    W3s[(e >> 8) * M3_LDH + (e & 255)] = w3r[u];
  }
  __syncthreads();
  store_tile(P.h2);

  // ---- head: wave w reduces k in [KW w, KW w + KW), KW = 256 / NW; output columns = lanes i < 16 ----
  constexpr int KW = M3_H / NW;
  f32x16 hacc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[rb][r] = 0.0f;
#pragma unroll
  for (int q = 0; q < KW / 8; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(W3s + (i & 15) * M3_LDH + KW * wave + 8 * q + 4 * hi);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Hs + (M3_R * rb + i) * M3_LDH + KW * wave + 8 * q + 4 * hi);
#pragma unroll
      for (int r = 0; r < 4; ++r) hacc[rb] = mfma32(a[r], b[r], hacc[rb]);
    }
  }
  if (i < M3_OMAX) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) Red[(wave * R + M3_R * rb + (r & 3) + 8 * (r >> 2) + 4 * hi) * M3_OMAX + i] = hacc[rb][r];
  }
  __syncthreads();
  for (int e = tid; e < R * M3_OMAX; e += NT) {
    const int row = e >> 4, o = e & 15;
    if (o < g.O && row0 + row < M) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; w += 4)                  // k-quarters (or -eighths) in wave order, four at a time
        v += (Red[((w + 0) * R + row) * M3_OMAX + o] + Red[((w + 1) * R + row) * M3_OMAX + o]) +
             (Red[((w + 2) * R + row) * M3_OMAX + o] + Red[((w + 3) * R + row) * M3_OMAX + o]);
      v += P.b3 ? P.b3[o] : 0.0f;
      P.y[(size_t)(row0 + row) * g.O + o] = m3_act(g.last_act, v);
    }
  }
}

// 32-row tiles only (launches that do not fill the chip otherwise): 128 registers, two workgroups per CU
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void mlp3_fwd_kernel(Mlp3Dev g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int net = blockIdx.x / g.tiles_small;
  mlp3_body<NW, 1>(g, lds, net, (blockIdx.x - net * g.tiles_small) * M3_R);
}

// Workgroups [0, n_big * tiles_big) run 64-row tiles of networks [0, n_big); the rest 32-row tiles of the other networks.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void mlp3_fwd_mixed_kernel(Mlp3Dev g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x, nb = g.n_big * g.tiles_big;
  if (b < nb) {
    const int net = b / g.tiles_big;
    mlp3_body<NW, 2>(g, lds, net, (b - net * g.tiles_big) * 2 * M3_R);
  } else {
    const int s = b - nb, net = s / g.tiles_small;
    mlp3_body<NW, 1>(g, lds, g.n_big + net, (s - net * g.tiles_small) * M3_R);
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
  static_assert(M3_H * M3_LDX <= M3_PAN && M3_OMAX * M3_LDH <= M3_PAN, "aliases fit the panel space");
  // 64-row tiles for as many networks as fill one round of the chip (one such workgroup per CU), 32-row tiles for the
  // rest (a second, half-length round); launches that do not even fill the chip with 32-row tiles keep those.
  // TRL_MLP3_BIG=<n> pins the number of networks on 64-row tiles (development).
  static const int pin = [] { const char* e = getenv("TRL_MLP3_BIG"); return e ? atoi(e) : -1; }();
  const int t_small = trl_ceil_div(M, M3_R), t_big = trl_ceil_div(M, 2 * M3_R);
  int n_big = 0;
  if (G * t_small > 256 && t_big <= 256) n_big = std::min(G, 256 / t_big);
  if (pin >= 0) n_big = std::min(G, pin);
  g.n_big = n_big; g.tiles_big = t_big; g.tiles_small = t_small;
  const int rmax = n_big > 0 ? 2 * M3_R : M3_R;
  const int lds = (rmax * M3_LDX + rmax * M3_LDH + M3_PAN + 8 * rmax * M3_OMAX) * (int)sizeof(float);
  static int attr_lds[2] = {0, 0};
  const int which = n_big > 0 ? 1 : 0;
  if (lds > attr_lds[which]) {
    const void* fn = which ? (const void*)mlp3_fwd_mixed_kernel<8> : (const void*)mlp3_fwd_kernel<8>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("mlp3_forward: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds[which] = lds;
  }
  // 8 waves = 8 column blocks of 32 per workgroup (the 4-wave instantiation measured slower: 34.5 / 82 us against 19.9 / 48.5)
  const dim3 grid(n_big * t_big + (G - n_big) * t_small);
  if (which) hipLaunchKernelGGL(mlp3_fwd_mixed_kernel<8>, grid, dim3(512), lds, (hipStream_t)stream, g);
  else       hipLaunchKernelGGL(mlp3_fwd_kernel<8>, grid, dim3(512), lds, (hipStream_t)stream, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
