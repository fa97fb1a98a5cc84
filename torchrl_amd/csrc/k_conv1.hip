// K16c -- direct kernels for the first conv layer of the Atari nets (CNNBase, torchrl/networks/base.py:59-107, on
// ScaledFloatFrame'd uint8 frames, env/atari_wrapper.py:230-240) when it has <= 16 output channels and <= 256 taps
// (config/dqn_pong.json: 4 x 8 x 8 -> 16).
//
// The generic implicit-GEMM kernel stages both operands through LDS and gives every wave a 32 x 32 quadrant: with
// 16 output channels half of each MFMA is padding and the staging is pure overhead, because a row of the virtual
// im2col matrix is only ever used by ONE wave.  Here a wave works on v_mfma_f32_16x16x4_f32 tiles, 16 channels
// wide, and reads the frames straight into the MFMA operand layout:
//   forward   lane (i, g) of a 16-position block loads the dword of taps 16u + 4g .. + 3 of position i -- its 4 bytes
//             are the A operands of 4 MFMA steps (reduction index of step (u, r), lane group g: k = 16u + 4g + r);
//             the weights of output channel j sit in 64 registers of lane (j, g) for the whole kernel.  No LDS, no
//             barrier; neighbouring positions are sw bytes apart, so a wave's load is contiguous.
//   weight gradient   reduction over positions (4 per MFMA step, one per lane group): lane (j, g) loads the dword of
//             taps 64q + 4j .. + 3 of position m0 + g; byte r feeds column block (q, r), whose column j therefore
//             IS tap 64q + 4j + r -- one dword load per 4 MFMAs.  A operand = dY * act'(Y), 64 consecutive floats
//             per step.  Waves reduce through LDS, workgroups through the fixed-order fold (deterministic).
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_conv.h"

#define C1_THREADS 256
#define C1_MAX_WG 512

__device__ __forceinline__ float c1_byte(uint32_t x, int r, float scale, float shift) {
  return fmaf((float)((x >> (8 * r)) & 0xffu), scale, shift);
}

// A SECOND problem of the same geometry in the same launch (DQN's target net on next_obs beside the online net on obs,
// dqn.py:38-52): workgroups [n_main, 2 n_main) run it -- one launch boundary less, and one ragged last round of
// workgroups instead of two (864 workgroups are 3.4 rounds of 256 CUs, 1728 are 6.75).
struct Conv1Second { const uint8_t* frames; const float* w; const float* bias; float* y; };

template <int KU>                                   // K / 16
__global__ __launch_bounds__(C1_THREADS) void conv1_fwd_direct_kernel(ConvSrc cv, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     int M, int K, int Cout, int act, PermJobs pj, int n_main,
                                                                     Conv1Second second) {
  const int n_work = second.frames ? 2 * n_main : n_main;
  if ((int)blockIdx.x >= n_work) {                   // riders: the later layers' weights into the reduction order of their kernels
    conv_perm_jobs(pj, blockIdx.x - n_work, gridDim.x - n_work, threadIdx.x, C1_THREADS);
    return;
  }
  int block = blockIdx.x;
  if (block >= n_main) { block -= n_main; cv.frames = second.frames; w = second.w; bias = second.bias; y = second.y; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  // byte offset of every 4-tap group, once per workgroup (two run-time integer divisions apiece: sixteen of them per lane
  // were ~600 of a wave's ~1500 vector instructions, next to 256 MFMAs)
  __shared__ uint32_t tap_tab[4 * KU];
  if (threadIdx.x < 4 * KU) tap_tab[threadIdx.x] = conv_tap_offset(cv, 4u * threadIdx.x);
  __syncthreads();
  const int mw = (block * (C1_THREADS / 64) + wave) * 64;          // this wave's 64 output positions
  if (mw >= M) return;
  f32x4 wreg[KU];
  uint32_t tap[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int kk = 16 * u + 4 * g;
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    wreg[u] = j < Cout ? *reinterpret_cast<const f32x4*>(w + (size_t)j * K + kk) : z;
    tap[u] = tap_tab[4 * u + g];
  }
  f32x4 acc[4];
  uint32_t dw[2][KU];
  auto load_block = [&](int rb, uint32_t (&d)[KU]) {
    const int m = min(mw + 16 * rb + j, M - 1);     // rows past M are computed on a clamped address and never stored
    const uint32_t ro = conv_row_offset(cv, (uint32_t)m);
    // (uniform base + 32-bit lane offset: one add per load instead of a 64-bit address built per lane; the batch is < 4 GiB)
#pragma unroll
    for (int u = 0; u < KU; ++u) d[u] = *reinterpret_cast<const uint32_t*>(cv.frames + (size_t)(uint32_t)(ro + tap[u]));
  };
  load_block(0, dw[0]);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    if (rb < 3) load_block(rb + 1, dw[(rb + 1) & 1]);              // next block's loads fly under this block's MFMAs
    f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const uint32_t x = dw[rb & 1][u];
#pragma unroll
      for (int r = 0; r < 4; ++r) a = mfma16(c1_byte(x, r, cv.scale, cv.shift), wreg[u][r], a);
    }
    acc[rb] = a;
  }
  if (j < Cout) {
    const float b = bias ? bias[j] : 0.0f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mw + 16 * rb + 4 * g + r;      // D register r of lane (j, g) is row 4g + r, column j
        float v = acc[rb][r] + b;
        if (act == TRL_ACT_TANH) v = trl_tanh(v);
        else if (act == TRL_ACT_RELU) v = fmaxf(v, 0.0f);
        if (m < M) y[(size_t)m * Cout + j] = v;
      }
  }
}

template <int KQ, int GATE>                         // K / 64; activation whose derivative gates dY
__global__ __launch_bounds__(C1_THREADS) void conv1_bwdw_direct_kernel(ConvSrc cv, const float* __restrict__ dy,
                                                                      const float* __restrict__ yg, float* __restrict__ part,
                                                                      float* __restrict__ colsum_part, int M, int K, int Cout,
                                                                      int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [4 waves][16][K] partial dW, then [4][16] partial db
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  uint32_t tapd[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) tapd[q] = conv_tap_offset(cv, (uint32_t)(64 * q + 4 * j));   // (4 per lane, once per workgroup's long loop)
  f32x4 acc[KQ][4];
#pragma unroll
  for (int q = 0; q < KQ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float csum = 0.0f;
  const int rows_per_wave = rows_per_wg / (C1_THREADS / 64);
  const int m_lo = blockIdx.x * rows_per_wg + wave * rows_per_wave, m_hi = min(M, m_lo + rows_per_wave);

  float a_cur = 0.0f, a_nxt = 0.0f;
  uint32_t d_cur[KQ], d_nxt[KQ];
  auto load_step = [&](int m0, float& a, uint32_t (&d)[KQ]) {
    const int m = m0 + g;
    const bool ok = m < m_hi;
    const int mc = ok ? m : m_lo;                   // a clamped row: its A operand is 0, so what B holds is irrelevant
    float dz = 0.0f;
    if (ok && j < Cout) {
      dz = dy[(size_t)m * Cout + j];
      if (GATE == TRL_ACT_TANH) { const float o = yg[(size_t)m * Cout + j]; dz *= 1.0f - o * o; }
      if (GATE == TRL_ACT_RELU) dz = yg[(size_t)m * Cout + j] > 0.0f ? dz : 0.0f;
    }
    a = dz;
    const uint8_t* p = cv.frames + conv_row_offset(cv, (uint32_t)mc);
#pragma unroll
    for (int q = 0; q < KQ; ++q) d[q] = *reinterpret_cast<const uint32_t*>(p + tapd[q]);
  };
  if (m_lo < m_hi) load_step(m_lo, a_cur, d_cur);
  for (int m0 = m_lo; m0 < m_hi; m0 += 4) {
    if (m0 + 4 < m_hi) load_step(m0 + 4, a_nxt, d_nxt);            // next step's loads fly under this step's MFMAs
    csum += a_cur;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = mfma16(a_cur, c1_byte(d_cur[q], r, cv.scale, cv.shift), acc[q][r]);
    a_cur = a_nxt;
#pragma unroll
    for (int q = 0; q < KQ; ++q) d_cur[q] = d_nxt[q];
  }
  // ---- waves -> LDS: D register rr of lane (j, g) in block (q, r) is dW[cout = 4g + rr][tap = 64q + 4j + r] ----
  float* mine = lds + wave * 16 * K;
#pragma unroll
  for (int q = 0; q < KQ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) mine[(4 * g + rr) * K + 64 * q + 4 * j + r] = acc[q][r][rr];
  csum += __shfl_xor(csum, 16, 64);
  csum += __shfl_xor(csum, 32, 64);                 // lanes with equal j now hold the wave's column sum of channel j
  float* cs = lds + 4 * 16 * K;
  if (lane < 16) cs[wave * 16 + lane] = csum;
  __syncthreads();
  for (int e = tid; e < Cout * K; e += C1_THREADS)
    part[(size_t)blockIdx.x * Cout * K + e] = (lds[e] + lds[16 * K + e]) + (lds[2 * 16 * K + e] + lds[3 * 16 * K + e]);
  if (colsum_part && tid < Cout)
    colsum_part[(size_t)blockIdx.x * Cout + tid] = (cs[tid] + cs[16 + tid]) + (cs[32 + tid] + cs[48 + tid]);
}

bool trl_conv1_direct_ok(int K, int Cout, const float* w) {     // w: the forward's weight matrix (16-byte loads), or null
  return Cout <= 16 && K <= 256 && (K & 63) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
}

int trl_conv1_direct_fwd(const ConvSrc& cv, const float* w, const float* bias, float* y, int M, int K, int Cout, int act,
                         const PermJobs& pj, hipStream_t stream, const uint8_t* frames2, const float* w2, const float* bias2,
                         float* y2) {
  const int n_main = trl_ceil_div(M, 64 * (C1_THREADS / 64));
  const Conv1Second second{frames2, w2, bias2, y2};
  const dim3 grid((frames2 ? 2 : 1) * n_main + ((pj.n > 0 || pj.n_dx > 0) ? CONV_PERM_BLOCKS : 0)), block(C1_THREADS);
  switch (K / 16) {
    case 4:  hipLaunchKernelGGL(conv1_fwd_direct_kernel<4>, grid, block, 0, stream, cv, w, bias, y, M, K, Cout, act, pj, n_main, second); break;
    case 8:  hipLaunchKernelGGL(conv1_fwd_direct_kernel<8>, grid, block, 0, stream, cv, w, bias, y, M, K, Cout, act, pj, n_main, second); break;
    case 12: hipLaunchKernelGGL(conv1_fwd_direct_kernel<12>, grid, block, 0, stream, cv, w, bias, y, M, K, Cout, act, pj, n_main, second); break;
    default: hipLaunchKernelGGL(conv1_fwd_direct_kernel<16>, grid, block, 0, stream, cv, w, bias, y, M, K, Cout, act, pj, n_main, second); break;
  }
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

static void bwdw_split(int M, int& n_wg, int& rows_per_wg) {
  rows_per_wg = trl_ceil_div(trl_ceil_div(M, C1_MAX_WG), 16) * 16;   // 4 waves x a multiple of 4 positions
  n_wg = trl_ceil_div(M, rows_per_wg);
}
int trl_conv1_direct_bwdw_workspace(int M, int K, int Cout) {
  int n_wg, rows;
  bwdw_split(M, n_wg, rows);
  return n_wg * (Cout * K + Cout);
}

template <int KQ>
static int launch_bwdw(const ConvSrc& cv, const float* dy, const float* yg, int gate, float* part, float* cpart, int M, int K,
                       int Cout, int n_wg, int rows, hipStream_t s) {
  const int lds = (int)sizeof(float) * (4 * 16 * K + 64);
  auto go = [&](auto kern) -> int {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 256);
      if (e != hipSuccess) { trl_set_error("conv1_bwdw: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C1_THREADS), lds, s, cv, dy, yg, part, cpart, M, K, Cout, rows);
    return TRL_OK;
  };
  int rc;
  if (gate == TRL_ACT_TANH) rc = go(conv1_bwdw_direct_kernel<KQ, TRL_ACT_TANH>);
  else if (gate == TRL_ACT_RELU) rc = go(conv1_bwdw_direct_kernel<KQ, TRL_ACT_RELU>);
  else rc = go(conv1_bwdw_direct_kernel<KQ, TRL_ACT_NONE>);
  if (rc) return rc;
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

int trl_conv1_direct_bwdw(const ConvSrc& cv, const float* dy, const float* y_gate, int gate_act, float* dw, float* db,
                          float* workspace, int M, int K, int Cout, hipStream_t stream) {
  int n_wg, rows;
  bwdw_split(M, n_wg, rows);
  float* part = workspace;
  float* cpart = db ? workspace + (size_t)n_wg * Cout * K : nullptr;
  const int gate = y_gate ? gate_act : TRL_ACT_NONE;
  int rc;
  switch (K / 64) {
    case 1:  rc = launch_bwdw<1>(cv, dy, y_gate, gate, part, cpart, M, K, Cout, n_wg, rows, stream); break;
    case 2:  rc = launch_bwdw<2>(cv, dy, y_gate, gate, part, cpart, M, K, Cout, n_wg, rows, stream); break;
    case 3:  rc = launch_bwdw<3>(cv, dy, y_gate, gate, part, cpart, M, K, Cout, n_wg, rows, stream); break;
    default: rc = launch_bwdw<4>(cv, dy, y_gate, gate, part, cpart, M, K, Cout, n_wg, rows, stream); break;
  }
  if (rc) return rc;
  return trl_fold_partials(part, dw, Cout * K, cpart, db, db ? Cout : 0, n_wg, stream);
}
