// K4 -- GAE / discounted-return as an LDS-staged, wave-parallel linear-recurrence scan.
//
// Both recurrences (torchrl/replay_buffers/on_policy.py:16-70) have the form
//     y_t = a_t + c_t * y_{t+1}
// GAE (filter f_t = 1 - tl_t or 1):  a = (r + (1-d) g V_{t+1} - V_t) f,  c = (1-d) g tau f,  y_T = 0
// discounted return, filtered:       a = r + tl V_t,  c = (1-d) g (1-tl),                    y_T = last_value
// discounted return, unfiltered:     a = r,           c = (1-d) g
// Affine maps compose associatively, so one wave scans a whole env: each lane
// folds a contiguous run of time steps, a 6-step shuffle suffix-scan composes the
// 64 lane maps, then each lane replays its run.  A workgroup owns ENV_TILE envs:
// coalesced (t, env) loads -> LDS [env][t] (odd stride, conflict free) -> scan
// along t -> coalesced stores.  Time is processed in chunks of T_CHUNK from the
// end with a per-env carry, so any T fits in LDS.
//
// HBM traffic: 16 B read + 8 B written per env-step (V_{t+1} re-reads hit L1/L2).
// Round 2 (cfg 2, 128 x 2048, rocprofv3): 10.1 -> 6.7 us = 0.94 TB/s of algorithmic traffic -- all of a thread's loads of a
// chunk in flight at once (one memory round trip instead of one per 16 time steps), no second read of V in the write-back,
// the wave's envs scanned in lockstep (their shuffle chains overlap), 8 envs per workgroup (256 workgroups at N = 2048
// instead of 128 on 256 CUs).  What is left is launch + three dependent phases (load, LDS scan, store) of a 6 MB kernel.
// Round 6: tiles are dealt to workgroups so that each XCD's L2 sees a contiguous run of envs (see below).
#include "trl_common.h"

#define ENV_TILE 8
#define T_CHUNK 256
#define GAE_THREADS 256

template <int MODE>  // 0 = GAE, 1 = discounted return
__global__ __launch_bounds__(GAE_THREADS) void gae_scan_kernel(
    const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ term,
    const float* __restrict__ tl, const float* __restrict__ last_v, const float* __restrict__ last_t, float* __restrict__ adv,
    float* __restrict__ ret, int T, int N, float gamma, float tau, int tl_filter) {
  constexpr int LDT = T_CHUNK + 1;
  __shared__ float s_a[ENV_TILE * LDT];
  __shared__ float s_c[ENV_TILE * LDT];
  __shared__ float s_carry[ENV_TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin (b -> XCD b % 8), and an env tile's row piece is
  // 32 B of a 128-B line -- with tile = b the four tiles sharing a line sat in four different L2s and every line was
  // fetched four times over the fabric (round 5: FETCH_SIZE 8.26 MB against 4.19 MB read).  Each XCD now takes a
  // CONTIGUOUS run of tiles (XCD x: its (G - x + 7) / 8 workgroups, in order), so a line is fetched into one L2.
  const int G = gridDim.x, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  int first = 0;
  for (int y = 0; y < xcd; ++y) first += (G - y + 7) >> 3;
  const int n0 = (first + q) * ENV_TILE;
  const int e_ld = tid % ENV_TILE, t_ld = tid / ENV_TILE;       // staging role
  const int n_ld = n0 + e_ld;

  __shared__ float s_lastv[ENV_TILE];
  if (tid < ENV_TILE) {
    float lv = 0.0f;
    if (n0 + tid < N) lv = last_v[n0 + tid] * (last_t ? 1.0f - last_t[n0 + tid] : 1.0f);
    s_lastv[tid] = lv;
    s_carry[tid] = (MODE == 0) ? 0.0f : lv;
  }

  for (int t_hi = T; t_hi > 0; t_hi -= T_CHUNK) {
    const int t_lo = max(0, t_hi - T_CHUNK);
    const int len = t_hi - t_lo;
    __syncthreads();
    // ---- stage coefficients ----
    // All of a thread's loads of the chunk are issued before the first one is consumed (fixed trip count, predicated):
    // the chunk then costs ONE memory round trip instead of one per 16 time steps (the 128 x 2048 scan of cfg 2 went
    // from 8 dependent round trips to 1).
    constexpr int TPT = T_CHUNK / (GAE_THREADS / ENV_TILE);        // time steps per thread and chunk (16)
    float lr[TPT], lv[TPT], ld[TPT], ltl[TPT], lvn[TPT];
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int tt = t_ld + q * (GAE_THREADS / ENV_TILE);
      const int t = t_lo + tt;
      const bool ok = tt < len && n_ld < N;
      const size_t i = ok ? (size_t)t * N + n_ld : 0;
      lr[q] = ok ? rew[i] : 0.0f;
      lv[q] = ok ? val[i] : 0.0f;
      ld[q] = ok ? term[i] : 0.0f;
      ltl[q] = (ok && tl_filter) ? tl[i] : 0.0f;
      lvn[q] = (MODE == 0 && ok && t + 1 < T) ? val[i + N] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int tt = t_ld + q * (GAE_THREADS / ENV_TILE);
      if (tt < len) {
        const int t = t_lo + tt;
        float a = 0.f, c = 0.f;
        if (n_ld < N) {
          const float r = lr[q], v = lv[q], nd = 1.0f - ld[q];
          const float tlv = ltl[q];
          if (MODE == 0) {
            const float vn = (t + 1 < T) ? lvn[q] : s_lastv[e_ld];
            const float f = 1.0f - tlv;
            a = (r + nd * gamma * vn - v) * f;
            c = nd * gamma * tau * f;
          } else {
            a = r + tlv * v;
            c = nd * gamma * (1.0f - tlv);
          }
        }
        s_a[e_ld * LDT + tt] = a;
        s_c[e_ld * LDT + tt] = c;
      }
    }
    __syncthreads();
    // ---- scan: wave w handles envs w, w+4, ... ; lane owns steps [lo, hi) of the chunk ----
    const int seg = (len + 63) >> 6;
    // The wave's ENV_TILE / 4 envs advance in lockstep: their lane-composition scans are independent chains, so the
    // cross-lane shuffles of one env travel under those of the others (12 dependent shuffles per env otherwise).
    constexpr int EPW = ENV_TILE / (GAE_THREADS / 64);
    const int lo = min(lane * seg, len), hi = min(lo + seg, len);
    float fa[EPW], fb[EPW];                       // y_lo = fa + fb * y_hi
#pragma unroll
    for (int u = 0; u < EPW; ++u) {
      const float* pa = s_a + (wave + u * (GAE_THREADS / 64)) * LDT;
      const float* pc = s_c + (wave + u * (GAE_THREADS / 64)) * LDT;
      fa[u] = 0.f; fb[u] = 1.f;
      for (int t = hi - 1; t >= lo; --t) { fa[u] = pa[t] + pc[t] * fa[u]; fb[u] = pc[t] * fb[u]; }
    }
    // inclusive suffix scan over lanes: S_l = F_l o F_{l+1} o ... o F_63
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float a2[EPW], b2[EPW];
#pragma unroll
      for (int u = 0; u < EPW; ++u) { a2[u] = __shfl_down(fa[u], o, 64); b2[u] = __shfl_down(fb[u], o, 64); }
      if (lane + o < 64) {
#pragma unroll
        for (int u = 0; u < EPW; ++u) { fa[u] = fa[u] + fb[u] * a2[u]; fb[u] = fb[u] * b2[u]; }
      }
    }
#pragma unroll
    for (int u = 0; u < EPW; ++u) {
      const int e = wave + u * (GAE_THREADS / 64);
      float* pa = s_a + e * LDT;
      const float* pc = s_c + e * LDT;
      const float carry = s_carry[e];
      const float a_nx = __shfl_down(fa[u], 1, 64), b_nx = __shfl_down(fb[u], 1, 64);
      float y = (lane == 63) ? carry : (a_nx + b_nx * carry);   // value entering this lane's run
      for (int t = hi - 1; t >= lo; --t) { y = pa[t] + pc[t] * y; pa[t] = y; }
      const float y0 = __shfl(fa[u], 0, 64) + __shfl(fb[u], 0, 64) * carry;
      if (lane == 0) s_carry[e] = y0;
    }
    __syncthreads();
    // ---- write back (values still in the staging registers: no second read of V) ----
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int tt = t_ld + q * (GAE_THREADS / ENV_TILE);
      if (tt < len && n_ld < N) {
        const size_t i = (size_t)(t_lo + tt) * N + n_ld;
        const float y = s_a[e_ld * LDT + tt], v = lv[q];
        if (MODE == 0) { adv[i] = y; ret[i] = y + v; }
        else           { ret[i] = y; adv[i] = y - v; }
      }
    }
  }
}

static int launch_scan(int mode, const float* rew, const float* val, const float* term, const float* tl,
                       const float* last_v, const float* last_t, float* adv, float* ret, int T, int N,
                       float gamma, float tau, int tl_filter, void* stream) {
  if (!rew || !val || !term || !last_v || !adv || !ret) { trl_set_error("gae: null pointer"); return TRL_EINVAL; }
  if (tl_filter && !tl) { trl_set_error("gae: tl_filter set but time_limits is null"); return TRL_EINVAL; }
  if (T < 0 || N < 0) { trl_set_error("gae: negative size"); return TRL_EINVAL; }
  if (T == 0 || N == 0) return TRL_OK;
  dim3 grid(trl_ceil_div(N, ENV_TILE)), block(GAE_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(gae_scan_kernel<0>, grid, block, 0, s, rew, val, term, tl, last_v, last_t, adv, ret, T, N, gamma, tau, tl_filter);
  else
    hipLaunchKernelGGL(gae_scan_kernel<1>, grid, block, 0, s, rew, val, term, tl, last_v, last_t, adv, ret, T, N, gamma, tau, tl_filter);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_gae_f32(const float* rewards, const float* values, const float* terminals,
                           const float* time_limits, const float* last_value, const float* last_terminal,
                           float* advs, float* rets, int T, int N, float gamma, float tau, int tl_filter,
                           void* stream) {
  return launch_scan(0, rewards, values, terminals, time_limits, last_value, last_terminal, advs, rets, T, N, gamma, tau, tl_filter, stream);
}

extern "C" int trl_discount_reward_f32(const float* rewards, const float* values, const float* terminals,
                                       const float* time_limits, const float* last_value,
                                       const float* last_terminal, float* advs, float* rets, int T, int N,
                                       float gamma, int tl_filter, void* stream) {
  return launch_scan(1, rewards, values, terminals, time_limits, last_value, last_terminal, advs, rets, T, N, gamma, 0.f, tl_filter, stream);
}
