// K1 + K2 + K3 -- the whole vectorised on-policy rollout as two launches: a persistent
// policy / env kernel that walks all T steps, then one streaming value pass over the T x N cells.
//
// Replaces n_steps x VecOnPolicyCollector.take_actions
// (torchrl/collector/on_policy.py:90-155; loop base.py:108-122) on the
// synthetic env: environments are independent, the networks are shared and
// read-only during collection, so a workgroup can carry 16 envs through all T
// steps with no inter-workgroup communication -- the 2 host<->device round
// trips and the python env loop per step of the reference disappear, and so do
// per-step kernel launches.
//
// The rollout is a latency problem (T strictly sequential steps, only N/16 independent tiles), so
// the sequential kernel carries only what the recurrence needs -- the policy and the env:
//   * workgroup = 4 waves = 16 envs on v_mfma_f32_16x16x4_f32; wave mo computes 16 of the 64 hidden
//     features; its slices of W1, W2, W3 and the env matrices stay in registers for the whole
//     rollout -- the per-step critical path has no weight fetches;
//   * per step: L1 5 MFMA -> tanh -> LDS exchange -> L2 16 MFMA -> tanh -> head partial 4 MFMA -> LDS
//     exchange -> action = tanh(mean + std*eps) for the two action dims the lane feeds to the env GEMM ->
//     env step as a 14-MFMA GEMM [obs, act] x [A; B] (every wave: everyone owns next_obs in the
//     operand layout of the next L1) -> reward / done -> partial reset from the Philox reset stream ->
//     ring-buffer row store;
//   * exploration noise (Philox4x32-10 + Box-Muller, ~300 instructions per block) is produced 16 steps
//     at a time, one (step, env) per lane, and parked in LDS -- 1/8 block per lane-step instead of 1.
// The value network is not part of the recurrence: V(obs) for every stored cell, and the over-length
// bootstrap reward + discount * V(next_obs) (on_policy.py:135-143), are computed afterwards by
// `value_pass_kernel` at full-chip parallelism (each wave owns whole 16-sample tiles, all weights in
// registers, layer outputs chain as the next layer's B operand -- no LDS, no barriers).
//
// HBM traffic per env-step: 176 B algorithmic (obs 68 + next_obs 68 + act 24 +
// value/reward/terminal/time_limit 16; the obs read is only at launch) + 4 B
// old_logp (+24 B if host noise is supplied) + 76 B re-read / marker traffic of the value pass.
#include <algorithm>
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_philox.h"

#define RO_THREADS 256
#define RO_ENVS 16
// development aid (tools/exp_rollout.py): per-phase cycle totals of workgroup 0 into the tail of ep_log
#ifdef TRL_EXP_CLK
#define CLK_DECL long long clk_prev = clock64(); float clk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CLK(ph) { const long long c_ = clock64(); clk_acc[ph] += (float)(c_ - clk_prev); clk_prev = c_; }
#else
#define CLK_DECL
#define CLK(ph)
#endif
#define RO_NB 16                                     // steps of exploration noise generated per batch

struct RolloutDev {
  const float *pf_params, *vf_params, *env_A, *env_B;
  float reward_scale; int horizon; int64_t env_seed_base;
  float* cur_obs; int32_t *t_env, *cur_step, *episode_idx; float* ep_return;
  const float* noise; int64_t noise_step0;
  float *obs, *next_obs, *acts, *values, *rewards, *terminals, *time_limits, *old_logp;
  int rows, top, N, n_steps, max_episode_frames; float discount;
  double* epoch_reward; int32_t* ep_count; float* ep_log; int ep_cap; int step0;
  int tanh_action, deterministic, store;
  // running observation normaliser (NORM instantiation only)
  double* norm_state; float* policy_obs; double* norm_ws; float norm_clip; int norm_update, norm_partial_reset;
  double* clear_hdr;          // 16 bytes zeroed by the launch (the next launch's header), or NULL
  float* boot;                // (N) V(next_obs) of the last stored step, written by the value pass, or NULL
  uint32_t* pub_dst; const uint32_t* pub_src; int64_t pub_words;     // copied to host memory by the value pass
  const uint32_t* noise_flag; uint32_t noise_stamp;                   // noise staged on another stream: wait for the stamp
  // workgroups [n_ro_wg, gridDim.x) stage the NEXT rollout's noise block (see trl_rollout_t.stage_*)
  int D, A;                   // runtime dims of the RT instantiations (template D / A are then capacities)
  int n_ro_wg; const f32x4* stg_src; unsigned long long* stg_dst; int64_t stg_n4;
  const uint32_t* stg_ready; uint32_t stg_job; uint32_t* stg_state; uint32_t* stg_ack;
};
#define RO_STAGERS 16

// ---- grid-wide exchange between the (always co-resident) rollout workgroups ----
// No ticket counter: a workgroup publishes its partials, waits for the stores to be acknowledged and then
// stamps its slot of the epoch table with the (launch-unique, monotone) step number; readers poll the
// stamps.  Everything travels through agent-scope relaxed atomics; the poll is capped so a scheduling
// accident raises the error word (ws header [2]) instead of hanging the GPU.
__device__ __forceinline__ float ro_row_sum16(float v) {          // sum over the 16 env lanes of a lane group
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
  return v;
}
#define NORM_SLOT 40                                 // doubles per workgroup partial: sum x [D] | sum x^2 [D] | any flag | .. | stamp

template <int D, int H, int A> struct RoShape {
  static_assert(H == 64 && D > 16 && D <= 32 && A <= 8, "instantiated for 16 < D <= 32, H == 64, A <= 8");
  // LDS: b1[H] | b2[H] | b3[8] | logstd[8] | H1 staging [H][TL] | headp [4][8][16] | eps [RO_NB][8][16]
  static constexpr int O_B1 = 0, O_B2 = H, O_B3 = 2 * H, O_LS = O_B3 + 8, O_H1 = O_LS + 8, O_HP = O_H1 + H * TL,
                       O_EPS = O_HP + 4 * 8 * 16, LDS_FLOATS = O_EPS + RO_NB * 8 * 16;
};

// Host block -> device buffer with device-scope stores (past the per-XCD L2: nothing to write back or invalidate when the
// stamp is published), four host reads in flight per thread; the last of `wgs` workgroups to finish resets the arrival
// counter and publishes the stamp (and, if asked, tells the host).
__device__ __forceinline__ void stage_block(const f32x4* __restrict__ src, unsigned long long* __restrict__ dst, int64_t n4,
                                            int wg, int wgs, int threads, uint32_t* __restrict__ state, uint32_t stamp,
                                            uint32_t* __restrict__ host_ack) {
  const int64_t stride = (int64_t)wgs * threads;
  for (int64_t e = (int64_t)wg * threads + threadIdx.x; e < n4; e += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (e + q * stride < n4) v[q] = src[e + q * stride];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (e + q * stride < n4) {
        const unsigned long long lo = ((unsigned long long)__float_as_uint(v[q][1]) << 32) | __float_as_uint(v[q][0]);
        const unsigned long long hi = ((unsigned long long)__float_as_uint(v[q][3]) << 32) | __float_as_uint(v[q][2]);
        __hip_atomic_store(dst + 2 * (e + q * stride), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2 * (e + q * stride) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's stores are acknowledged ...
  __syncthreads();                                             // ... and so are the workgroup's
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == (unsigned)wgs - 1) {                         // the last workgroup to finish publishes the stamp
      __hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(state, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (host_ack) __hip_atomic_store(host_ack, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// RT = false: the dims are the template's (the benchmark shape: 17 observations, 6 actions).
// RT = true: D and A are CAPACITIES (17 or 32 features, 8 actions) and the actual dims come with the launch (a.D, a.A):
// missing features / actions are masked to zero where they are loaded and never stored, parameter offsets and row strides
// follow the actual dims -- Hopper- (11 / 3), Swimmer-, Walker-shaped synthetic tasks collect in the same ONE launch
// (torchrl/collector/on_policy.py:90-155 and torchrl/networks/base.py:8-44 are shape-generic).  D = 32 (WIDE, a.D in
// [18, 32] -- Ant's 27 observations): features 16..31 are a second k group of the first layer and a second row block of
// the env GEMM instead of the single 17th feature that rides on the VALU.
template <int D, int H, int A, int ACT, bool NORM, bool RT = false>
__global__ __launch_bounds__(RO_THREADS, 1) void rollout_kernel(RolloutDev a) {
  using S = RoShape<D, H, A>;
  constexpr bool WIDE = D > 17;
  static_assert(!WIDE || RT, "the wide tile exists as a runtime-dims instantiation only");
  static_assert(!(RT && NORM), "the cooperative (normalised) rollout is instantiated for the benchmark shape");
  const int Dr = RT ? a.D : D, Ar = RT ? a.A : A;   // actual dims (row strides of obs / acts, parameter offsets)
  // offsets inside the flat parameter block for the actual dims (MlpFlat, trl_mlp.h)
  const int F_W1 = 0, F_B1 = H * Dr, F_W2 = F_B1 + H, F_B2 = F_W2 + H * H, F_W3 = F_B2 + H, F_B3 = F_W3 + Ar * H,
            F_LS = F_B3 + Ar;
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_FLOATS];
  if (!NORM && (int)blockIdx.x >= a.n_ro_wg) {
    // ---- a stager: the NEXT rollout's noise block, host -> device, next to this rollout's own workgroups ----
    __shared__ int s_go;
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();            // 100 MHz
      int go = 1;
      while (__hip_atomic_load(a.stg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.stg_job) {
        __builtin_amdgcn_s_sleep(64);
        // 30 us.  A host that runs ahead of the device published the block long before this launch started; one that does
        // not is still drawing it (~1 ms), and the rollout must not wait for that -- the stagers leave, the host sees no
        // acknowledgement and stages the block itself under the update (trl_stage_h2d_f32).
        if (wall_clock64() - t0 > 3000ull) { go = 0; break; }
      }
      s_go = go;
    }
    __syncthreads();
    if (s_go)
      stage_block(a.stg_src, a.stg_dst, a.stg_n4, (int)blockIdx.x - a.n_ro_wg, (int)gridDim.x - a.n_ro_wg, RO_THREADS,
                  a.stg_state, a.stg_job, a.stg_ack);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int mo = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it stay scalar
  const int j = lane & 15, g = lane >> 4, i = j;
  const int n = blockIdx.x * RO_ENVS + j;
  const bool valid = n < a.N;
  const float* gp = a.pf_params;
  const bool has_lo = g < Ar, has_hi = 4 + g < Ar;          // the lane's two action dims: g and 4 + g
  const int o_lo = has_lo ? g : 0, o_hi = has_hi ? 4 + g : 0;
  if (blockIdx.x == 0 && tid == 0 && a.clear_hdr) { a.clear_hdr[0] = 0.0; a.clear_hdr[1] = 0.0; }   // the NEXT launch's header
  if (a.noise_flag && tid == 0) {
    // The noise block is staged by a kernel on another stream (trl_stage_h2d_f32), normally long before this launch
    // starts -- then its lines cannot be in this XCD's L2 (invalidated at the launch boundary, not read since).  If the
    // stamp is not there yet, wait for it and then drop whatever the wait may have pulled in.
    unsigned spins = 0;
    while (__hip_atomic_load(a.noise_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.noise_stamp) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > (1u << 24)) {                              // ~seconds: the staging launch never ran
        if (blockIdx.x == 0 && a.epoch_reward) a.epoch_reward[0] = __builtin_nan("");
        break;
      }
    }
    if (spins) __threadfence();
  }

  // ---- one-time setup: biases / logstd to LDS, this wave's weight slices to registers ----
  for (int e = tid; e < H; e += RO_THREADS) { lds[S::O_B1 + e] = gp[F_B1 + e]; lds[S::O_B2 + e] = gp[F_B2 + e]; }
  if (tid < 8) {
    lds[S::O_B3 + tid] = tid < Ar ? gp[F_B3 + (tid < Ar ? tid : 0)] : 0.0f;
    lds[S::O_LS + tid] = tid < Ar ? gp[F_LS + (tid < Ar ? tid : 0)] : 0.0f;
  }
  // which of the tile's input features exist: feature 4g + q of the lane's x operand, feature 16 + g (narrow tile) or
  // 16 + 4g + q (wide tile); a missing feature is loaded from an in-range address and replaced by zero
  bool fv[4], fv2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { fv[q] = 4 * g + q < Dr; fv2[q] = WIDE && 16 + 4 * g + q < Dr; }
  const bool f16 = !WIDE && 16 + g < Dr;             // narrow tile: the 5th k step carries feature 16 + g
  const bool row_f = i < Dr, row_f2 = WIDE && 16 + i < Dr;   // env GEMM rows (next features i, 16 + i) that exist
  // A operands, lane (i, g): the k index of MFMA step (slice sl, r) is feature 16 sl + 4 g + r
  // XK: obs k steps of the lane (4 for features 4g + q, then 1 (narrow: feature 16 + g) or 4 (wide: 16 + 4g + q))
  constexpr int XK = WIDE ? 8 : 5, EK = XK + 2;      // EK: k steps of the env GEMM (obs steps + the two action steps)
  float w1r[XK], w2r[4][4], w3a[4], we0[EK], w16[EK], we1[WIDE ? EK : 1];
  {
    const int row = 16 * mo + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) w1r[r] = fv[r] ? gp[F_W1 + row * Dr + (fv[r] ? 4 * g + r : 0)] : 0.0f;
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) w1r[4 + r] = fv2[r] ? gp[F_W1 + row * Dr + (fv2[r] ? 16 + 4 * g + r : 0)] : 0.0f;
    } else {
      w1r[4] = f16 ? gp[F_W1 + row * Dr + (f16 ? 16 + g : 0)] : 0.0f;
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
      for (int r = 0; r < 4; ++r) w2r[sl][r] = gp[F_W2 + row * H + 16 * sl + 4 * g + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) w3a[r] = (i < Ar) ? gp[F_W3 + (i < Ar ? i : 0) * H + 16 * mo + 4 * g + r] : 0.0f;
    // env GEMM next^T[f][j] = sum_k M[f][k] [obs; act]^T[k][j] for features f = 0..15 (we0, MFMA A operand of
    // row f = i) and, wide tile, f = 16..31 (we1); narrow tile: feature 16 is a per-lane partial dot over the lane's own
    // inputs (w16) + a lane-group sum
    static_assert(WIDE || D == 17, "the narrow env step keeps exactly one feature outside the 16-row MFMA tile");
    const bool has16 = !WIDE && Dr > 16;
#pragma unroll
    for (int q = 0; q < EK; ++q) {
      // k of step q: obs feature 4g + q (q < 4); narrow: obs feature 16 + g (q == 4); wide: obs feature 16 + 4g + (q - 4)
      // (4 <= q < 8); then action g and action 4 + g
      const float* mat = q < XK ? a.env_A : a.env_B;
      int k; bool kin;
      if (q < 4) { k = 4 * g + q; kin = k < Dr; }
      else if (q < XK) { k = WIDE ? 16 + 4 * g + (q - 4) : 16 + g; kin = k < Dr; }
      else { k = (q == XK) ? g : 4 + g; kin = k < Ar; }
      const int kc = kin ? k : 0;
      we0[q] = (kin && row_f) ? mat[kc * Dr + (row_f ? i : 0)] : 0.0f;
      w16[q] = (kin && has16) ? mat[kc * Dr + (has16 ? 16 : 0)] : 0.0f;
      if constexpr (WIDE) we1[q] = (kin && row_f2) ? mat[kc * Dr + (row_f2 ? 16 + i : 0)] : 0.0f;
    }
  }
  __syncthreads();

  float* S_H1 = lds + S::O_H1;
  float* headp = lds + S::O_HP;
  float* S_EPS = lds + S::O_EPS;
  const float* b1s = lds + S::O_B1 + 16 * mo + 4 * g;
  const float* b2s = lds + S::O_B2 + 16 * mo + 4 * g;

  // the lane's two action dims
  const float ls_lo = fminf(fmaxf(lds[S::O_LS + o_lo], -20.0f), 2.0f), ls_hi = fminf(fmaxf(lds[S::O_LS + o_hi], -20.0f), 2.0f);
  const float std_lo = __expf(ls_lo), std_hi = __expf(ls_hi);
  const float iv_lo = __expf(-2.0f * ls_lo), iv_hi = __expf(-2.0f * ls_hi);
  const float b3_lo = lds[S::O_B3 + o_lo], b3_hi = lds[S::O_B3 + o_hi];

  // ---- per-env state (replicated in all 4 waves and all 4 lane groups) ----
  // x operand: lane (env j, g) holds obs features 4g..4g+3 and (g == 0) feature 16
  float xb[XK];
#pragma unroll
  for (int r = 0; r < 4; ++r) xb[r] = (valid && fv[r]) ? a.cur_obs[(size_t)n * Dr + (fv[r] ? 4 * g + r : 0)] : 0.0f;
  if constexpr (WIDE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[4 + r] = (valid && fv2[r]) ? a.cur_obs[(size_t)n * Dr + (fv2[r] ? 16 + 4 * g + r : 0)] : 0.0f;
  } else {
    xb[4] = (valid && f16) ? a.cur_obs[(size_t)n * Dr + (f16 ? 16 + g : 0)] : 0.0f;
  }
  // NORM: xb is the env's RAW state (it drives the dynamics), xp what the policy sees (normalised, or raw right
  // after a partial reset -- the reference's behaviour); without a normaliser they are the same.
  float xp[XK];
  __shared__ double s_mean[NORM ? D : 1], s_var[NORM ? D : 1], s_sum[NORM ? 2 * D + 1 : 1];
  __shared__ double s_cnt;
  unsigned* nhdr = NORM ? reinterpret_cast<unsigned*>(a.norm_ws) : nullptr;
  if constexpr (NORM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xp[r] = valid ? a.policy_obs[(size_t)n * D + 4 * g + r] : 0.0f;
    xp[4] = (valid && g == 0) ? a.policy_obs[(size_t)n * D + 16] : 0.0f;
    if (tid < D) { s_mean[tid] = a.norm_state[tid]; s_var[tid] = a.norm_state[D + tid]; }
    if (tid == 0) s_cnt = a.norm_state[2 * D];
    __syncthreads();
  } else {
#pragma unroll
    for (int q = 0; q < XK; ++q) xp[q] = xb[q];
  }
  int t_env = valid ? a.t_env[n] : 0;
  int cur_step = valid ? a.cur_step[n] : 0;
  int ep_idx = valid ? a.episode_idx[n] : 0;
  float ep_ret = valid ? a.ep_return[n] : 0.0f;
  const int64_t env_seed = a.env_seed_base + n;
  double rew_sum = 0.0;
  const bool dev_noise = !a.deterministic && !a.noise;
  CLK_DECL

  for (int t = 0; t < a.n_steps; ++t) {
    const int row = (a.top + t) % a.rows;
    const size_t cell = (size_t)row * a.N + n;

    // ---- exploration noise: host stream (reference parity, distribution.py:67-70) or device Philox ----
    if (dev_noise && (t % RO_NB) == 0) {
      // lane (env j, g) of wave mo draws both Philox blocks (8 normals) of step t + 4 mo + g
      const int ts = 4 * mo + g;
      const int64_t gs = a.noise_step0 + t + ts;
      float z0[4], z1[4];
      philox_normals4((uint32_t)(gs & 0xFFFFFFFFll), (uint32_t)((gs >> 32) & 0xFFFFFFFFll), 0u, TRL_TAG_NOISE, env_seed, z0);
      if (Ar > 4)
        philox_normals4((uint32_t)(gs & 0xFFFFFFFFll), (uint32_t)((gs >> 32) & 0xFFFFFFFFll), 1u, TRL_TAG_NOISE, env_seed, z1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        S_EPS[(ts * 8 + c) * 16 + j] = z0[c];
        if (4 + c < Ar) S_EPS[(ts * 8 + 4 + c) * 16 + j] = z1[c];
      }
      __syncthreads();   // readers of the previous batch finished before the barriers of the step that ended it
    }
    float eps_lo = 0.0f, eps_hi = 0.0f;
    if (dev_noise) {
      eps_lo = S_EPS[((t % RO_NB) * 8 + o_lo) * 16 + j];
      eps_hi = S_EPS[((t % RO_NB) * 8 + o_hi) * 16 + j];
    } else if (a.noise) {
      eps_lo = valid ? a.noise[((size_t)t * a.N + n) * Ar + o_lo] : 0.0f;
      eps_hi = valid ? a.noise[((size_t)t * a.N + n) * Ar + o_hi] : 0.0f;
    }

    CLK(0)
    // ---- policy forward ----
    f32x4 h1 = *reinterpret_cast<const f32x4*>(b1s);
#pragma unroll
    for (int q = 0; q < XK; ++q) h1 = mfma16(w1r[q], xp[q], h1);
    // the observation part of the env step does not wait for the action: it runs in the shadow of the barriers
    f32x4 e0 = f32x4{0.f, 0.f, 0.f, 0.f}, e1 = f32x4{0.f, 0.f, 0.f, 0.f};
    float p16 = 0.0f;
#pragma unroll
    for (int q = 0; q < XK; ++q) {
      e0 = mfma16(we0[q], xb[q], e0);
      if constexpr (WIDE) e1 = mfma16(we1[q], xb[q], e1);
      else p16 = fmaf(w16[q], xb[q], p16);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[r] = act_fn<ACT>(h1[r]);
    store_T(S_H1, mo, h1, j, g);
    CLK(1)
    __syncthreads();
    CLK(2)
    f32x4 h2 = *reinterpret_cast<const f32x4*>(b2s), h2b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 4; sl += 2) {                     // two interleaved accumulation chains
      const f32x4 b0 = load_T(S_H1, sl, j, g), b1 = load_T(S_H1, sl + 1, j, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { h2 = mfma16(w2r[sl][r], b0[r], h2); h2b = mfma16(w2r[sl + 1][r], b1[r], h2b); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[r] = act_fn<ACT>(h2[r] + h2b[r]);
    f32x4 hp = f32x4{0.f, 0.f, 0.f, 0.f};                  // partial head over the own 16 features: D[o][env]
#pragma unroll
    for (int r = 0; r < 4; ++r) hp = mfma16(w3a[r], h2[r], hp);
    if (g < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) headp[(mo * 8 + 4 * g + r) * 16 + j] = hp[r];
    }
    CLK(3)
    __syncthreads();
    CLK(4)
    float mean_lo = b3_lo, mean_hi = b3_hi;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mean_lo += headp[(w * 8 + o_lo) * 16 + j]; mean_hi += headp[(w * 8 + o_hi) * 16 + j]; }

    // action + log-prob under the collecting policy (continuous_policy.py:123-129, distribution.py:33-45)
    const float z_lo = fmaf(std_lo, eps_lo, mean_lo), z_hi = fmaf(std_hi, eps_hi, mean_hi);
    const float act_lo = has_lo ? (a.tanh_action ? trl_tanh(z_lo) : z_lo) : 0.0f;
    const float act_hi = has_hi ? (a.tanh_action ? trl_tanh(z_hi) : z_hi) : 0.0f;
    // ---- env step: next^T[f][env] = sum_k M[f][k] [obs; act]^T[k][env] (action part) ----
    e0 = mfma16(we0[XK], act_lo, e0);
    e0 = mfma16(we0[XK + 1], act_hi, e0);
    if constexpr (WIDE) {
      e1 = mfma16(we1[XK], act_lo, e1);
      e1 = mfma16(we1[XK + 1], act_hi, e1);
    } else {
      p16 = fmaf(w16[XK], act_lo, p16);
      p16 = fmaf(w16[XK + 1], act_hi, p16);
    }
    // lane-group sums (same env, g = 0..3) as ones x value MFMAs: every lane gets the total
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const float e16 = WIDE ? 0.0f : mfma16(1.0f, p16, zero4)[0];
    const float act_sq = mfma16(1.0f, fmaf(act_lo, act_lo, act_hi * act_hi), zero4)[0];   // all action dims
    float logp = 0.0f;
    if (mo == 3) {                                          // only the wave that stores old_logp needs it
      float zc, lp = 0.0f;
      if (has_lo) lp += gauss_logp_term(act_lo, mean_lo, iv_lo, ls_lo, a.tanh_action, zc);
      if (has_hi) lp += gauss_logp_term(act_hi, mean_hi, iv_hi, ls_hi, a.tanh_action, zc);
      logp = mfma16(1.0f, lp, zero4)[0];
    }
    CLK(5)
    float nx[XK];
#pragma unroll
    for (int r = 0; r < 4; ++r) nx[r] = trl_tanh(e0[r]);     // features 4g + r (absent ones: tanh(0) = 0)
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) nx[4 + r] = trl_tanh(e1[r]);   // features 16 + 4g + r
    } else {
      nx[4] = (g == 0 && Dr > 16) ? trl_tanh(e16) : 0.0f;    // feature 16
    }

    CLK(6)
    // feature 0 lives in lane group 0 (r == 0): broadcast to the other groups with a selector MFMA
    const float nx0 = mfma16(1.0f, g == 0 ? nx[0] : 0.0f, zero4)[0];
    const float raw_rew = a.reward_scale * (nx0 - 0.1f * act_sq);
    t_env += 1; cur_step += 1;
    const bool done = t_env >= a.horizon;
    const bool surpass = cur_step >= a.max_episode_frames;
    // ---- NormObs.observation(next_obs) (base_wrapper.py:116-119): statistics over ALL envs, then the filter ----
    float nxn[XK];                                          // what is stored as next_obs / fed to the policy
    bool any_reset = false;
    if constexpr (NORM) {
      if (a.norm_update) {
        double* part = a.norm_ws + 2 + (size_t)((t & 1) * gridDim.x + blockIdx.x) * NORM_SLOT;
        if (mo == 0) {                                        // every wave holds the same 16 envs: wave 0 publishes
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const float v = (valid && (q < 4 || g == 0)) ? nx[q] : 0.0f;
            const float s1 = ro_row_sum16(v), s2 = ro_row_sum16(v * v);
            const int f = q < 4 ? 4 * g + q : 16;
            if (j == 0 && (q < 4 || g == 0)) {
              __hip_atomic_store(part + f, (double)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(part + D + f, (double)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          const bool fl = __ballot(valid && (done || surpass)) != 0ull;
          if (lane == 0) __hip_atomic_store(part + 2 * D, fl ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partials acknowledged before the stamp goes out
          const unsigned long long stamp = (unsigned long long)(a.noise_step0 + t + 1);
          if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + NORM_SLOT - 1), stamp, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          // wave 0 polls every workgroup's stamp (lanes stride over the workgroups)
          for (int w = lane; w < (int)gridDim.x; w += 64) {
            const unsigned long long* st = reinterpret_cast<const unsigned long long*>(
                a.norm_ws + 2 + (size_t)((t & 1) * gridDim.x + w) * NORM_SLOT + NORM_SLOT - 1);
            int it = 0;
            while (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp) {
              __builtin_amdgcn_s_sleep(1);
              if (++it > (1 << 22)) { __hip_atomic_store(nhdr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
          }
        }
        __syncthreads();
        if (tid < 2 * D + 1) {                                // fixed-order fold over the workgroups, 8 loads in flight
          const double* col = a.norm_ws + 2 + (size_t)((t & 1) * gridDim.x) * NORM_SLOT + tid;
          double acc = 0.0;
          int w = 0;
          for (; w + 8 <= (int)gridDim.x; w += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __hip_atomic_load(col + (size_t)(w + k) * NORM_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
          }
          for (; w < (int)gridDim.x; ++w) acc += __hip_atomic_load(col + (size_t)w * NORM_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_sum[tid] = acc;
        }
        __syncthreads();
        if (tid < D) {                                        // Chan merge (update_mean_var_count, :44-60), same in every workgroup
          const double bn = (double)a.N, cnt = s_cnt;
          const double bmean = s_sum[tid] / bn, bvar = fmax(s_sum[D + tid] / bn - bmean * bmean, 0.0);
          const double delta = bmean - s_mean[tid], tot = cnt + bn;
          const double m2 = s_var[tid] * cnt + bvar * bn + delta * delta * cnt * bn / tot;
          s_mean[tid] += delta * bn / tot;
          s_var[tid] = m2 / tot;
        }
        __syncthreads();
        if (tid == 0) s_cnt += (double)a.N;
        any_reset = s_sum[2 * D] != 0.0;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int f = q < 4 ? 4 * g + q : 16;
        const double z = ((double)nx[q] - s_mean[f]) / (sqrt(s_var[f]) + 1e-4);
        nxn[q] = (q < 4 || g == 0) ? (float)fmin(fmax(z, -(double)a.norm_clip), (double)a.norm_clip) : 0.0f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < XK; ++q) nxn[q] = nx[q];
    }
    ep_ret += raw_rew;
    if (mo == 0 && g == 0 && valid) {
      rew_sum += (double)raw_rew;
      if (done) {                                           // on_policy.py:128-130
        const int slot = atomicAdd(a.ep_count, 1);
        if (slot < a.ep_cap) { a.ep_log[slot * 3 + 0] = (float)(a.step0 + t); a.ep_log[slot * 3 + 1] = (float)n; a.ep_log[slot * 3 + 2] = ep_ret; }
      }
    }
    if (done) ep_ret = 0.0f;

    float xn[XK];
#pragma unroll
    for (int q = 0; q < XK; ++q) xn[q] = nx[q];
    if (done || surpass) {                                  // partial_reset (vecenv.py:47-51) + counters (:148)
      ep_idx += 1; t_env = 0; cur_step = 0;
      float z[4];
      philox_normals4((uint32_t)ep_idx, 0u, (uint32_t)g, TRL_TAG_RESET, env_seed, z);     // features 4g..4g+3
#pragma unroll
      for (int c = 0; c < 4; ++c) xn[c] = fv[c] ? z[c] : 0.0f;
      if constexpr (WIDE) {
        philox_normals4((uint32_t)ep_idx, 0u, (uint32_t)(4 + g), TRL_TAG_RESET, env_seed, z);   // features 16 + 4g ..
#pragma unroll
        for (int c = 0; c < 4; ++c) xn[4 + c] = fv2[c] ? z[c] : 0.0f;
      } else {
        xn[4] = 0.0f;
        if (g == 0 && Dr > 16) {
          philox_normals4((uint32_t)ep_idx, 0u, 4u, TRL_TAG_RESET, env_seed, z);          // feature 16
          xn[4] = z[0];
        }
      }
    }

    // ---- ring-buffer row `row` (replay_buffers/base.py:19-29), one key group per wave ----
    if (valid && a.store) {
      if (mo == 0 || mo == 1) {
        float* dst = (mo == 0 ? a.obs : a.next_obs) + cell * Dr;
        const float* src = mo == 0 ? xp : nxn;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!RT || fv[r]) dst[4 * g + r] = src[r];
        if constexpr (WIDE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (fv2[r]) dst[16 + 4 * g + r] = src[4 + r];
        } else {
          if (g == 0 && (!RT || Dr > 16)) dst[16] = src[4];
        }
      } else if (mo == 2) {
        if (has_lo) a.acts[cell * Ar + g] = act_lo;
        if (has_hi) a.acts[cell * Ar + 4 + g] = act_hi;
      } else {
        // one store instruction for the four per-cell scalars: lane group g writes key g.  values[cell]
        // carries the over-length marker to the value pass, which overwrites it with V(obs)
        float* dst = g == 0 ? a.values : g == 1 ? a.rewards : g == 2 ? a.terminals : a.time_limits;
        const float val = g == 0 ? (surpass ? 1.0f : 0.0f) : g == 1 ? raw_rew
                        : g == 2 ? ((done || surpass) ? 1.0f : 0.0f) : (done ? 1.0f : 0.0f);
        dst[cell] = val;
        if (g == 0 && a.old_logp) a.old_logp[cell] = logp;
      }
    }
    if constexpr (NORM) {
      // partial_reset hands the collector the RAW observations of ALL envs whenever any env was reset
      // (base_wrapper.py:23-26, vecenv.py:47-51); otherwise the policy sees the normalised next_obs
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        float alt = xn[q];
        if (a.norm_partial_reset) {
          const int f = q < 4 ? 4 * g + q : 16;
          const double z = ((double)xn[q] - s_mean[f]) / (sqrt(s_var[f]) + 1e-4);
          alt = (q < 4 || g == 0) ? (float)fmin(fmax(z, -(double)a.norm_clip), (double)a.norm_clip) : 0.0f;
        }
        xp[q] = any_reset ? alt : nxn[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < XK; ++q) xp[q] = xn[q];
    }
#pragma unroll
    for (int q = 0; q < XK; ++q) xb[q] = xn[q];
    CLK(7)
  }
  if constexpr (NORM) {
    if (mo == 0 && valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.policy_obs[(size_t)n * D + 4 * g + r] = xp[r];
      if (g == 0) a.policy_obs[(size_t)n * D + 16] = xp[4];
    }
    if (blockIdx.x == 0 && a.norm_update) {
      if (tid < D) { a.norm_state[tid] = s_mean[tid]; a.norm_state[D + tid] = s_var[tid]; }
      if (tid == 0) a.norm_state[2 * D] = s_cnt;
    }
  }
#ifdef TRL_EXP_CLK
  if (blockIdx.x == 0 && lane == 0)
    for (int ph = 0; ph < 8; ++ph) a.ep_log[(a.ep_cap - 16) * 3 + mo * 8 + ph] = clk_acc[ph];
#endif

  // ---- persist env / collector state ----
  if (mo == 0 && valid) {
    float* dst = a.cur_obs + (size_t)n * Dr;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (!RT || fv[r]) dst[4 * g + r] = xb[r];
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (fv2[r]) dst[16 + 4 * g + r] = xb[4 + r];
    }
    if (g == 0) {
      if (!WIDE && (!RT || Dr > 16)) dst[16] = xb[4];
      a.t_env[n] = t_env; a.cur_step[n] = cur_step; a.episode_idx[n] = ep_idx; a.ep_return[n] = ep_ret;
    }
  }
  if (mo == 0) {
    const double tot = wave_sum(rew_sum);
    if (lane == 0 && a.epoch_reward) atomicAdd(a.epoch_reward, tot);
  }
}

// ---------------------------------------------------------------- value pass over the stored cells
// values[cell] = V(obs[cell]); rewards[cell] += discount * V(next_obs[cell]) where the rollout left
// the over-length marker (on_policy.py:135-143).  Sample m = t * N + n lives in ring cell
// ((top + t) % rows) * N + n.  One wave = one 16-sample tile at a time, all of W1 / W2 in registers
// (A operands), the D tile of layer l is the B operand of layer l + 1, the 1-wide head is 16 FMAs +
// two cross-lane adds.  84 MFMAs per tile, no LDS traffic besides the biases.
#define VP_THREADS 256

struct ValueDev {
  const float* vf_params; const float *obs, *next_obs; float *values, *rewards;
  int rows, top, N, n_steps; float discount;
  float* boot;                // (N) or NULL: V(next_obs) of the last step's row
  uint32_t* pub_dst; const uint32_t* pub_src; int64_t pub_words;     // epoch header + episode-log head -> page-locked host memory
  int D;                      // runtime input size of the RT instantiations
};

// RT: D is a capacity (17 or 32 input features), the actual size comes with the launch (see rollout_kernel)
template <int D, int H, int ACT, bool RT = false>
__global__ __launch_bounds__(VP_THREADS, RT ? 1 : 2) void value_pass_kernel(ValueDev a) {
  static_assert(H == 64 && D > 16 && D <= 32, "instantiated for 16 < D <= 32, H == 64");
  constexpr bool WIDE = D > 17;
  static_assert(!WIDE || RT, "the wide tile exists as a runtime-dims instantiation only");
  constexpr int XK = WIDE ? 8 : 5;
  const int Dr = RT ? a.D : D;
  const int F_W1 = 0, F_B1 = H * Dr, F_W2 = F_B1 + H, F_B2 = F_W2 + H * H, F_W3 = F_B2 + H, F_B3 = F_W3 + H;   // MlpFlat<Dr, H, 1>
  __shared__ __attribute__((aligned(16))) float sb[2 * H];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4, i = j;
  const float* gp = a.vf_params;
  for (int e = tid; e < H; e += VP_THREADS) { sb[e] = gp[F_B1 + e]; sb[H + e] = gp[F_B2 + e]; }
  bool fv[4], fv2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { fv[q] = 4 * g + q < Dr; fv2[q] = WIDE && 16 + 4 * g + q < Dr; }
  const bool f16 = !WIDE && 16 + g < Dr;
  const bool w2_al = !RT || ((reinterpret_cast<uintptr_t>(gp + F_W2) & 15) == 0);   // (odd D: the block is only 4-byte aligned)
  float w1[4][XK], w2[4][4][4], w3[4][4];
#pragma unroll
  for (int so = 0; so < 4; ++so) {
    const int row = 16 * so + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) w1[so][r] = fv[r] ? gp[F_W1 + row * Dr + (fv[r] ? 4 * g + r : 0)] : 0.0f;
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) w1[so][4 + r] = fv2[r] ? gp[F_W1 + row * Dr + (fv2[r] ? 16 + 4 * g + r : 0)] : 0.0f;
    } else {
      w1[so][4] = f16 ? gp[F_W1 + row * Dr + (f16 ? 16 + g : 0)] : 0.0f;
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      f32x4 w;
      if (w2_al) w = *reinterpret_cast<const f32x4*>(gp + F_W2 + row * H + 16 * sl + 4 * g);
      else __builtin_memcpy(&w, gp + F_W2 + row * H + 16 * sl + 4 * g, 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) w2[so][sl][r] = w[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) w3[so][r] = gp[F_W3 + 16 * so + 4 * g + r];
  }
  const float b3 = gp[F_B3];
  __syncthreads();

  // a tile's network input (raw loads; nothing waits on them here)
  auto fetch = [&](const float* src, size_t cell, bool ok, float (&xb)[XK]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[r] = (ok && fv[r]) ? src[cell * Dr + (fv[r] ? 4 * g + r : 0)] : 0.0f;
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[4 + r] = (ok && fv2[r]) ? src[cell * Dr + (fv2[r] ? 16 + 4 * g + r : 0)] : 0.0f;
    } else {
      xb[4] = (ok && f16) ? src[cell * Dr + (f16 ? 16 + g : 0)] : 0.0f;
    }
  };
  auto forward = [&](const float (&xb)[XK]) -> float {
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int so = 0; so < 4; ++so) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(sb + 16 * so + 4 * g);
#pragma unroll
      for (int q = 0; q < XK; ++q) acc = mfma16(w1[so][q], xb[q], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = act_fn<ACT>(acc[r]);
      h1[so] = acc;
    }
#pragma unroll
    for (int so = 0; so < 4; ++so) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(sb + H + 16 * so + 4 * g);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(w2[so][sl][r], h1[sl][r], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = act_fn<ACT>(acc[r]);
      h2[so] = acc;
    }
    float v = 0.0f;
#pragma unroll
    for (int so = 0; so < 4; ++so)
#pragma unroll
      for (int r = 0; r < 4; ++r) v = fmaf(w3[so][r], h2[so][r], v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v + b3;
  };

  // (posted writes into host memory, issued before the tiles: they complete under the pass)
  for (int64_t e = (int64_t)blockIdx.x * VP_THREADS + tid; e < a.pub_words; e += (int64_t)gridDim.x * VP_THREADS)
    a.pub_dst[e] = a.pub_src[e];
  const int64_t M = (int64_t)a.n_steps * a.N;
  const int64_t n_tiles = (M + 15) / 16;
  // The inputs of a wave's NEXT tile (observation slices and the bootstrap marker) are requested before the current tile's
  // 84 MFMAs, so no tile starts with a memory round trip.  Same values, same arithmetic; measured 41.2 -> 40.3 us at cfg 2
  // (the second wave of the SIMD already covered most of that round trip: the pass is bound by its tanh / MFMA issue).
  auto locate = [&](int64_t tile, bool& ok, int& t, int& n) -> size_t {
    const int64_t m = tile * 16 + j;
    ok = tile < n_tiles && m < M;
    t = ok ? (int)(m / a.N) : 0;
    n = ok ? (int)(m - (int64_t)t * a.N) : 0;
    return (size_t)((a.top + t) % a.rows) * a.N + n;
  };
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  bool ok, ok_n;
  int t, n, t_n, n_n;
  float xb[XK], xn[XK];
  size_t cell = locate(tile, ok, t, n), cell_n;
  fetch(a.obs, cell, ok, xb);
  float marker = ok ? a.values[cell] : 0.0f, marker_n;
  for (; tile < n_tiles; tile += stride) {
    cell_n = locate(tile + stride, ok_n, t_n, n_n);
    fetch(a.obs, cell_n, ok_n, xn);
    marker_n = ok_n ? a.values[cell_n] : 0.0f;
    const float v = forward(xb);
    if (ok && g == 0) a.values[cell] = v;
    const bool last = ok && a.boot && t == a.n_steps - 1;              // the row the epoch's bootstrap value comes from
    if (__ballot(marker != 0.0f || last) != 0ull) {
      float x2[XK];
      fetch(a.next_obs, cell, ok, x2);
      const float v2 = forward(x2);
      if (ok && g == 0 && marker != 0.0f) a.rewards[cell] += a.discount * v2;
      if (last && g == 0) a.boot[n] = v2;
    }
#pragma unroll
    for (int q = 0; q < XK; ++q) xb[q] = xn[q];
    cell = cell_n; ok = ok_n; t = t_n; n = n_n; marker = marker_n;
  }
}

// Envs the normalised (cooperative) rollout can carry: its workgroups rendezvous once per step, so all of them
// must be resident at the same time.
template <int D, int H, int A, int ACT>
static int rollout_norm_capacity() {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rollout_kernel<D, H, A, ACT, true>, RO_THREADS, 0) != hipSuccess)
    return 0;
  return cus * per_cu * RO_ENVS;
}

template <int D, int H, int A, int ACT, bool RT = false>
static int launch_rollout(const RolloutDev& d, hipStream_t s, hipEvent_t value_wait = nullptr) {
  const int n_wg = trl_ceil_div(d.N, RO_ENVS);
  if constexpr (RT) {
    if (d.norm_state) { trl_set_error("rollout: the normalised rollout is instantiated for the benchmark shape only"); return TRL_EUNSUPPORTED; }
    RolloutDev e = d;
    e.n_ro_wg = n_wg;
    hipLaunchKernelGGL((rollout_kernel<D, H, A, ACT, false, true>), dim3(n_wg + (e.stg_n4 ? RO_STAGERS : 0)), dim3(RO_THREADS), 0, s, e);
  } else if (d.norm_state) {
    const int cap = rollout_norm_capacity<D, H, A, ACT>();
    if (d.norm_update && d.N > cap) {
      trl_set_error("rollout: %d envs with a running normaliser exceed the %d that can be co-resident", d.N, cap);
      return TRL_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((rollout_kernel<D, H, A, ACT, true>), dim3(n_wg), dim3(RO_THREADS), 0, s, d);
  } else {
    RolloutDev e = d;
    e.n_ro_wg = n_wg;
    hipLaunchKernelGGL((rollout_kernel<D, H, A, ACT, false>), dim3(n_wg + (e.stg_n4 ? RO_STAGERS : 0)), dim3(RO_THREADS), 0, s, e);
  }
  TRL_LAUNCH_CHECK();
  if (d.store) {
    if (value_wait) {                               // the value function's parameters are still being stepped on another stream
      hipError_t e = hipStreamWaitEvent(s, value_wait, 0);
      if (e != hipSuccess) { trl_set_error("rollout: hipStreamWaitEvent(value_wait_event): %s", hipGetErrorString(e)); return (int)e; }
    }
    ValueDev v{d.vf_params, d.obs, d.next_obs, d.values, d.rewards, d.rows, d.top, d.N, d.n_steps, d.discount, d.boot,
               d.pub_dst, d.pub_src, d.pub_words, d.D};
    const int64_t n_tiles = ((int64_t)d.n_steps * d.N + 15) / 16;
    const int grid = (int)(n_tiles / 4 + 1 < 512 ? n_tiles / 4 + 1 : 512);
    hipLaunchKernelGGL((value_pass_kernel<D, H, ACT, RT>), dim3(grid), dim3(VP_THREADS), 0, s, v);
    TRL_LAUNCH_CHECK();
  }
  return TRL_OK;
}

// ---- page-locked host block -> device buffer by a kernel, for a stream of its own ----
__global__ __launch_bounds__(256) void stage_h2d_kernel(const f32x4* __restrict__ src, unsigned long long* __restrict__ dst,
                                                        int64_t n4, uint32_t* __restrict__ state, uint32_t stamp) {
  stage_block(src, dst, n4, (int)blockIdx.x, (int)gridDim.x, 256, state, stamp, nullptr);
}
extern "C" int trl_stage_h2d_f32(const float* host_src, float* dev_dst, int64_t n, uint32_t* state, uint32_t stamp,
                                 void* stream) {
  TRL_REQUIRE(n >= 0 && (n & 3) == 0, "stage_h2d: n must be a multiple of 4");
  TRL_REQUIRE(state, "stage_h2d: null state");
  TRL_REQUIRE(!n || (host_src && dev_dst && ((reinterpret_cast<uintptr_t>(host_src) | reinterpret_cast<uintptr_t>(dev_dst)) & 15) == 0),
              "stage_h2d: null / misaligned pointer");
  // few workgroups: the transfer is paced by the host link (~50 GB/s), and they share the chip with the other stream's work
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(16, (n / 4 + 1023) / 1024));
  hipLaunchKernelGGL(stage_h2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(host_src),
                     reinterpret_cast<unsigned long long*>(dev_dst), n / 4, state, stamp);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// Shapes the persistent rollout carries without a normaliser: the benchmark shape (compile-time instantiation) and any
// 64-wide two-layer pair with 2..32 inputs and 1..8 actions (runtime-dims instantiations) -- what the fused update
// kernels carry (trl_ppo_partial_stride), so such a task is 1 rollout launch + 2 launches per minibatch.
extern "C" int trl_rollout_supported(int D, int H, int A, int act) {
  return H == 64 && D >= 2 && D <= 32 && A >= 1 && A <= 8 && (act == TRL_ACT_TANH || act == TRL_ACT_RELU);
}

extern "C" int trl_rollout_synth_f32(const trl_rollout_t* p, void* stream) {
  if (!p) { trl_set_error("rollout: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(p->pf_params && p->vf_params && p->env_A && p->env_B, "null network / env pointer");
  TRL_REQUIRE(p->cur_obs && p->t_env && p->cur_step && p->episode_idx && p->ep_return, "null env state pointer");
  const bool all_ring = p->obs && p->next_obs && p->acts && p->values && p->rewards && p->terminals && p->time_limits;
  const bool no_ring = !p->obs && !p->next_obs && !p->acts && !p->values && !p->rewards && !p->terminals && !p->time_limits;
  TRL_REQUIRE(all_ring || no_ring, "ring tensors must be all set or all NULL");
  TRL_REQUIRE(p->ep_count && p->ep_log && p->ep_cap >= 0, "null episode log");
  TRL_REQUIRE(p->N > 0 && p->n_steps >= 0 && p->rows > 0, "bad sizes");
  TRL_REQUIRE(p->top >= 0 && p->top < p->rows, "top outside ring");
  TRL_REQUIRE(no_ring || p->n_steps <= p->rows, "n_steps exceeds ring rows");
  TRL_REQUIRE(p->horizon > 0 && p->max_episode_frames > 0, "horizon / max_episode_frames must be positive");
  if (p->n_steps == 0) return TRL_OK;
  RolloutDev d;
  d.pf_params = p->pf_params; d.vf_params = p->vf_params; d.env_A = p->env_A; d.env_B = p->env_B;
  d.reward_scale = p->reward_scale; d.horizon = p->horizon; d.env_seed_base = p->env_seed_base;
  d.cur_obs = p->cur_obs; d.t_env = p->t_env; d.cur_step = p->cur_step; d.episode_idx = p->episode_idx;
  d.ep_return = p->ep_return; d.noise = p->noise; d.noise_step0 = p->noise_step0;
  d.obs = p->obs; d.next_obs = p->next_obs; d.acts = p->acts; d.values = p->values; d.rewards = p->rewards;
  d.terminals = p->terminals; d.time_limits = p->time_limits; d.old_logp = p->old_logp;
  d.rows = p->rows; d.top = p->top; d.N = p->N; d.n_steps = p->n_steps;
  d.max_episode_frames = p->max_episode_frames; d.discount = p->discount;
  d.epoch_reward = p->epoch_reward; d.ep_count = p->ep_count; d.ep_log = p->ep_log; d.ep_cap = p->ep_cap;
  d.step0 = p->step0; d.tanh_action = p->tanh_action; d.deterministic = p->deterministic; d.store = all_ring ? 1 : 0;
  d.norm_state = p->norm_state; d.policy_obs = p->policy_obs; d.norm_ws = p->norm_workspace; d.norm_clip = p->norm_clip;
  d.norm_update = p->norm_update; d.norm_partial_reset = p->normalize_partial_reset;
  d.clear_hdr = p->clear_header; d.boot = all_ring ? p->boot_values : nullptr;
  TRL_REQUIRE(p->publish_words >= 0 && (!p->publish_words || (p->publish_dst && p->publish_src && all_ring)) &&
              ((reinterpret_cast<uintptr_t>(p->publish_dst) | reinterpret_cast<uintptr_t>(p->publish_src)) & 3) == 0,
              "publish: needs ring tensors, non-null 4-byte aligned pointers");
  d.pub_dst = (uint32_t*)p->publish_dst; d.pub_src = (const uint32_t*)p->publish_src; d.pub_words = p->publish_words;
  d.noise_flag = p->noise ? p->noise_flag : nullptr; d.noise_stamp = p->noise_stamp;
  d.n_ro_wg = 0; d.stg_n4 = 0;
  if (p->stage_n) {
    TRL_REQUIRE(p->stage_n > 0 && (p->stage_n & 3) == 0 && p->stage_src && p->stage_dst && p->stage_ready && p->stage_state &&
                p->stage_ack && !p->norm_state, "stage: n % 4 == 0, non-null pointers, no observation normaliser");
    TRL_REQUIRE(((reinterpret_cast<uintptr_t>(p->stage_src) | reinterpret_cast<uintptr_t>(p->stage_dst)) & 15) == 0,
                "stage: 16-byte aligned blocks");
    d.stg_src = reinterpret_cast<const f32x4*>(p->stage_src); d.stg_dst = reinterpret_cast<unsigned long long*>(p->stage_dst);
    d.stg_n4 = p->stage_n / 4; d.stg_ready = p->stage_ready; d.stg_job = p->stage_job; d.stg_state = p->stage_state;
    d.stg_ack = p->stage_ack;
  }
  TRL_REQUIRE(!p->clear_header || (p->clear_header != p->epoch_reward && (void*)p->clear_header != (void*)p->ep_count),
              "clear_header must not be the header this launch accumulates into");
  TRL_REQUIRE(!p->norm_state || (p->policy_obs && p->norm_workspace), "normaliser needs policy_obs and its workspace");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t vw = (hipEvent_t)p->value_wait_event;
  d.D = p->D; d.A = p->A;
  if (p->D == 17 && p->H == 64 && p->A == 6) {
    if (p->act == TRL_ACT_TANH) return launch_rollout<17, 64, 6, TRL_ACT_TANH>(d, s, vw);
    if (p->act == TRL_ACT_RELU) return launch_rollout<17, 64, 6, TRL_ACT_RELU>(d, s, vw);
  }
  if (trl_rollout_supported(p->D, p->H, p->A, p->act) && !p->norm_state) {        // runtime-dims instantiations
    if (p->D <= 17) {
      if (p->act == TRL_ACT_TANH) return launch_rollout<17, 64, 8, TRL_ACT_TANH, true>(d, s, vw);
      return launch_rollout<17, 64, 8, TRL_ACT_RELU, true>(d, s, vw);
    }
    if (p->act == TRL_ACT_TANH) return launch_rollout<32, 64, 8, TRL_ACT_TANH, true>(d, s, vw);
    return launch_rollout<32, 64, 8, TRL_ACT_RELU, true>(d, s, vw);
  }
  trl_set_error("rollout: shape D=%d H=%d A=%d act=%d not instantiated", p->D, p->H, p->A, p->act);
  return TRL_EUNSUPPORTED;
}

extern "C" int trl_rollout_norm_workspace(int N) {
  if (N <= 0) { trl_set_error("trl_rollout_norm_workspace: N must be positive"); return TRL_EINVAL; }
  return 2 + 2 * trl_ceil_div(N, RO_ENVS) * NORM_SLOT;                 // doubles: header + two parities of partials
}
extern "C" int trl_rollout_norm_max_envs(int D, int H, int A, int act) {
  if (D == 17 && H == 64 && A == 6) {
    if (act == TRL_ACT_TANH) return rollout_norm_capacity<17, 64, 6, TRL_ACT_TANH>();
    if (act == TRL_ACT_RELU) return rollout_norm_capacity<17, 64, 6, TRL_ACT_RELU>();
  }
  trl_set_error("rollout: shape D=%d H=%d A=%d act=%d not instantiated", D, H, A, act);
  return TRL_EUNSUPPORTED;
}

// ---------------------------------------------------------------- env (re)start
__global__ __launch_bounds__(256) void synth_reset_kernel(float* __restrict__ cur_obs, int32_t* __restrict__ t_env,
                                                          int32_t* __restrict__ cur_step,
                                                          int32_t* __restrict__ episode_idx,
                                                          float* __restrict__ ep_return,
                                                          const uint8_t* __restrict__ mask, int N, int D,
                                                          int64_t seed_base) {
  const int nblk = (D + 3) / 4;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (env, 4-feature block)
  const int n = e / nblk, b = e - n * nblk;
  if (n >= N || (mask && !mask[n])) return;
  const int ep = episode_idx[n] + 1;
  float z[4];
  philox_normals4((uint32_t)ep, 0u, (uint32_t)b, TRL_TAG_RESET, seed_base + n, z);
  for (int c = 0; c < 4; ++c) if (4 * b + c < D) cur_obs[(size_t)n * D + 4 * b + c] = z[c];
  if (b == 0) {
    t_env[n] = 0;
    if (!mask) { cur_step[n] = 0; ep_return[n] = 0.0f; }      // collector-side state: full reset only
  }
}
__global__ __launch_bounds__(256) void synth_bump_episode_kernel(int32_t* __restrict__ episode_idx,
                                                                 const uint8_t* __restrict__ mask, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && (!mask || mask[n])) episode_idx[n] += 1;
}

extern "C" int trl_synth_reset_f32(float* cur_obs, int32_t* t_env, int32_t* cur_step, int32_t* episode_idx,
                                   float* ep_return, const uint8_t* mask, int N, int D, int64_t env_seed_base,
                                   void* stream) {
  TRL_REQUIRE(cur_obs && t_env && cur_step && episode_idx && ep_return, "null pointer");
  TRL_REQUIRE(N >= 0 && D > 0, "bad sizes");
  if (N == 0) return TRL_OK;
  const int nblk = (D + 3) / 4;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(synth_reset_kernel, dim3(trl_ceil_div((int64_t)N * nblk, 256)), dim3(256), 0, s, cur_obs,
                     t_env, cur_step, episode_idx, ep_return, mask, N, D, env_seed_base);
  TRL_LAUNCH_CHECK();
  // episode_idx is bumped in a second launch: threads of one env may sit in different workgroups
  hipLaunchKernelGGL(synth_bump_episode_kernel, dim3(trl_ceil_div(N, 256)), dim3(256), 0, s, episode_idx, mask, N);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- diagonal-Gaussian / TanhNormal log-prob
__global__ __launch_bounds__(256) void gauss_logp_kernel(const float* __restrict__ mean, const float* __restrict__ acts,
                                                         const float* __restrict__ logstd, float* __restrict__ out,
                                                         int B, int A, int tanh_action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float lp = 0.0f;
  for (int o = 0; o < A; ++o) {
    const float ls = fminf(fmaxf(logstd[o], -20.0f), 2.0f);
    float zc;
    lp += gauss_logp_term(acts[(size_t)b * A + o], mean[(size_t)b * A + o], __expf(-2.0f * ls), ls, tanh_action, zc);
  }
  out[b] = lp;
}

extern "C" int trl_gauss_logp_f32(const float* mean, const float* acts, const float* logstd, float* out, int B,
                                  int A, int tanh_action, void* stream) {
  TRL_REQUIRE(B >= 0 && A > 0, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(mean && acts && logstd && out, "null pointer");
  hipLaunchKernelGGL(gauss_logp_kernel, dim3(trl_ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, mean, acts,
                     logstd, out, B, A, tanh_action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- per-step on-policy collection
// The pieces of VecOnPolicyCollector.take_actions (torchrl/collector/on_policy.py:90-155) as stand-alone
// kernels, for environments the persistent rollout kernel cannot carry: with a running observation
// normaliser (NormObs) every step needs batch statistics over ALL envs before the next policy forward.

// pf.explore + log-prob for a state-independent-std Gaussian policy (continuous_policy.py:123-129, 180-188)
__global__ __launch_bounds__(256) void gauss_explore_kernel(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                            const float* __restrict__ eps, float* __restrict__ act,
                                                            float* __restrict__ logp, int N, int A, int tanh_action) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float lp = 0.0f;
  for (int o = 0; o < A; ++o) {
    const float ls = fminf(fmaxf(logstd[o], -20.0f), 2.0f);
    const float m = mean[(size_t)n * A + o];
    const float z = fmaf(__expf(ls), eps ? eps[(size_t)n * A + o] : 0.0f, m);
    const float a = tanh_action ? trl_tanh(z) : z;
    act[(size_t)n * A + o] = a;
    float zc;
    lp += gauss_logp_term(a, m, __expf(-2.0f * ls), ls, tanh_action, zc);
  }
  if (logp) logp[n] = lp;
}

extern "C" int trl_gauss_explore_f32(const float* mean, const float* logstd, const float* eps, float* act, float* logp,
                                     int N, int A, int tanh_action, void* stream) {
  TRL_REQUIRE(N >= 0 && A > 0, "bad sizes");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(mean && logstd && act, "null pointer");
  hipLaunchKernelGGL(gauss_explore_kernel, dim3(trl_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, mean, logstd,
                     eps, act, logp, N, A, tanh_action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// bookkeeping after env.step (on_policy.py:124-148): epoch reward, running returns (logged and cleared on done),
// over-length bootstrap r += discount * V(next_obs) * surpass, terminals = done | surpass = reset mask, and a
// device-side "any env flagged" word for the caller's choice of the next policy input (partial_reset, :145-147)
__global__ __launch_bounds__(256) void onpolicy_bookkeep_kernel(float* __restrict__ rew, const float* __restrict__ done,
                                                                const float* __restrict__ v_next, float discount,
                                                                float* __restrict__ terminals, int32_t* __restrict__ cur_step,
                                                                float* __restrict__ ep_return, int max_frames,
                                                                uint8_t* __restrict__ mask, int32_t* __restrict__ any_flag,
                                                                double* __restrict__ epoch_reward, int32_t* __restrict__ ep_count,
                                                                float* __restrict__ ep_log, int ep_cap, int step, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  double r = 0.0;
  bool flag = false;
  if (n < N) {
    const float raw = rew[n];
    r = (double)raw;
    const bool d = done[n] != 0.0f;
    const int cs = cur_step[n] + 1;
    float er = ep_return[n] + raw;
    if (d) {
      const int slot = atomicAdd(ep_count, 1);
      if (slot < ep_cap) { ep_log[slot * 3 + 0] = (float)step; ep_log[slot * 3 + 1] = (float)n; ep_log[slot * 3 + 2] = er; }
      er = 0.0f;
    }
    const bool surpass = cs >= max_frames;
    flag = d || surpass;
    rew[n] = raw + discount * v_next[n] * (surpass ? 1.0f : 0.0f);
    terminals[n] = flag ? 1.0f : 0.0f;
    cur_step[n] = flag ? 0 : cs;
    ep_return[n] = er;
    mask[n] = flag ? 1 : 0;
  }
  const double tot = wave_sum(r);
  if ((threadIdx.x & 63) == 0 && epoch_reward && tot != 0.0) atomicAdd(epoch_reward, tot);
  if (__ballot(flag) != 0ull && (threadIdx.x & 63) == 0) atomicOr(any_flag, 1);
}

extern "C" int trl_onpolicy_bookkeep_f32(float* rewards, const float* dones, const float* v_next, float discount,
                                         float* terminals, int32_t* cur_step, float* ep_return, int max_episode_frames,
                                         uint8_t* reset_mask, int32_t* any_flag, double* epoch_reward, int32_t* ep_count,
                                         float* ep_log, int ep_cap, int step, int N, void* stream) {
  TRL_REQUIRE(N >= 0 && ep_cap >= 0, "bad sizes");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(rewards && dones && v_next && terminals && cur_step && ep_return && reset_mask && any_flag && ep_count && ep_log,
              "null pointer");
  hipLaunchKernelGGL(onpolicy_bookkeep_kernel, dim3(trl_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, rewards,
                     dones, v_next, discount, terminals, cur_step, ep_return, max_episode_frames, reset_mask, any_flag,
                     epoch_reward, ep_count, ep_log, ep_cap, step, N);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// out = (*flag != 0) ? a : b, elementwise -- the flag stays on the device (no host round trip per step)
__global__ __launch_bounds__(256) void select_on_flag_kernel(const int32_t* __restrict__ flag, const float* __restrict__ a,
                                                             const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  const bool f = *flag != 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) out[e] = f ? a[e] : b[e];
}

extern "C" int trl_select_on_flag_f32(const int32_t* flag, const float* a, const float* b, float* out, int64_t n,
                                      void* stream) {
  TRL_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(flag && a && b && out, "null pointer");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(select_on_flag_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, flag, a, b, out, n);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// out = any(mask[0..N)) ? a : b, elementwise -- the off-policy collector's form of the same whole-array return
// (torchrl/collector/base.py:220-224): every block scans the (few-KB) reset mask itself, no flag word, no host sync
__global__ __launch_bounds__(256) void select_on_mask_kernel(const uint8_t* __restrict__ mask, int N, const float* __restrict__ a,
                                                             const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  int any = 0;
  for (int e = threadIdx.x; e < N; e += 256) any |= mask[e];
  const bool f = __syncthreads_or(any) != 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) out[e] = f ? a[e] : b[e];
}

extern "C" int trl_select_on_mask_f32(const uint8_t* mask, int N, const float* a, const float* b, float* out, int64_t n,
                                      void* stream) {
  TRL_REQUIRE(n >= 0 && N >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(mask && a && b && out, "null pointer");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(select_on_mask_kernel, dim3((unsigned)(blocks > 256 ? 256 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, mask, N, a, b, out, n);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
