// K1 + K2 + K3 -- the whole vectorised on-policy rollout as ONE persistent launch.
//
// Replaces n_steps x VecOnPolicyCollector.take_actions
// (torchrl/collector/on_policy.py:90-155; loop base.py:108-122) on the
// synthetic env: environments are independent, the networks are shared and
// read-only during collection, so a workgroup can carry 32 envs through all T
// steps with no inter-workgroup communication -- the 2 host<->device round
// trips and the python env loop per step of the reference disappear, and so do
// per-step kernel launches.
//
// Workgroup = 4 waves = 32 envs.  Wave (net, mo): net 0 = policy, 1 = value;
// mo = which 32 of the 64 hidden features it computes.  Per step and wave:
//   L1 9 MFMA -> tanh -> LDS exchange of H1 halves -> L2 32 MFMA -> tanh ->
//   partial head on VALU -> LDS exchange -> action = tanh(mean + std*eps) ->
//   env step as a 12-MFMA GEMM [obs, act] x [A; B] (computed by all 4 waves so
//   everyone owns next_obs in registers) -> reward / done / over-length
//   bootstrap (extra vf pass only when some env of the tile needs it) ->
//   partial reset from the Philox reset stream -> ring-buffer row store.
// The env output tile has the same register layout as the L1 input operand
// (see trl_mlp.h), so next_obs feeds the next step without touching memory.
//
// HBM traffic per env-step: 176 B algorithmic (obs 68 + next_obs 68 + act 24 +
// value/reward/terminal/time_limit 16; the obs read is only at launch) + 4 B
// old_logp (+24 B if host noise is supplied).
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_philox.h"

#define RO_THREADS 256

struct RolloutDev {
  const float *pf_params, *vf_params, *env_A, *env_B;
  float reward_scale; int horizon; int64_t env_seed_base;
  float* cur_obs; int32_t *t_env, *cur_step, *episode_idx; float* ep_return;
  const float* noise; int64_t noise_step0;
  float *obs, *next_obs, *acts, *values, *rewards, *terminals, *time_limits, *old_logp;
  int rows, top, N, n_steps, max_episode_frames; float discount;
  double* epoch_reward; int32_t* ep_count; float* ep_log; int ep_cap; int step0;
  int tanh_action, deterministic, store;
};

template <int D, int H, int A> struct RoShape {
  using LP = MlpLds<D, H, A>;
  using LV = MlpLds<D, H, 1>;
  static constexpr int ELD = ((D + A) % 2 == 0) ? D + A + 1 : D + A;   // odd stride
  static constexpr int OFF_PF = 0;
  static constexpr int OFF_VF = align4(LP::SIZE);
  static constexpr int OFF_ENV = OFF_VF + align4(LV::SIZE);
  static constexpr int OFF_XCH = OFF_ENV + align4(32 * ELD);
  static constexpr int OFF_HEAD = OFF_XCH + 2 * H * TRL_TLD;
  static constexpr int LDS_FLOATS = align4(OFF_HEAD + 2 * 2 * 8 * 32);
};

template <int D, int H, int A, int ACT>
__global__ __launch_bounds__(RO_THREADS, 1) void rollout_kernel(RolloutDev a) {
  using S = RoShape<D, H, A>;
  using LP = typename S::LP;
  using LV = typename S::LV;
  constexpr int NT = H / 32, KS = ksteps_for(D), KA = (A + 1) / 2, ELD = S::ELD;
  static_assert(NT == 2, "rollout kernel is laid out for H == 64 (2 feature tiles x 2 nets = 4 waves)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sp_pf = lds + S::OFF_PF;
  float* sp_vf = lds + S::OFF_VF;
  float* senv = lds + S::OFF_ENV;
  float* xch = lds + S::OFF_XCH;
  float* headp = lds + S::OFF_HEAD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = wave >> 1, mo = wave & 1;
  const int i = lane & 31, j = i, hi = lane >> 5;
  const int n = blockIdx.x * 32 + j;
  const bool valid = n < a.N;

  LP::load(sp_pf, a.pf_params, true, tid, RO_THREADS);
  LV::load(sp_vf, a.vf_params, false, tid, RO_THREADS);
  for (int e = tid; e < 32 * ELD; e += RO_THREADS) {
    const int f = e / ELD, k = e - f * ELD;
    float v = 0.0f;
    if (f < D) { if (k < D) v = a.env_A[k * D + f]; else if (k < D + A) v = a.env_B[(k - D) * D + f]; }
    senv[e] = v;
  }
  __syncthreads();

  const float* sp = net == 0 ? sp_pf : sp_vf;           // W1..B2 offsets are identical in LP and LV
  float* my_xch = xch + net * H * TRL_TLD;

  float stdv[A], lsv[A], inv_var[A];
#pragma unroll
  for (int o = 0; o < A; ++o) {
    lsv[o] = fminf(fmaxf(sp_pf[LP::LS + o], -20.0f), 2.0f);
    stdv[o] = __expf(lsv[o]);
    inv_var[o] = __expf(-2.0f * lsv[o]);
  }

  // ---- per-env state (replicated in all 4 waves and both lane halves) ----
  float xb[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) { const int k = rowmap(q, hi); xb[q] = (valid && k < D) ? a.cur_obs[(size_t)n * D + k] : 0.0f; }
  int t_env = valid ? a.t_env[n] : 0;
  int cur_step = valid ? a.cur_step[n] : 0;
  int ep_idx = valid ? a.episode_idx[n] : 0;
  float ep_ret = valid ? a.ep_return[n] : 0.0f;
  const int64_t env_seed = a.env_seed_base + n;
  double rew_sum = 0.0;

  // both networks' forward; every wave returns the full policy mean and the value
  auto forward = [&](const float (&x)[KS], float (&mean)[A], float& value) {
    f32x16 h1 = act_tile<ACT>(layer1_tile<D, LP::LD1, KS>(bias_tile(sp + LP::B1 + 32 * mo, hi), sp + LP::W1, mo, x, i, hi));
#pragma unroll
    for (int r = 0; r < 16; ++r) my_xch[(32 * mo + rowmap(r, hi)) * TRL_TLD + j] = h1[r];
    __syncthreads();
    f32x16 hh[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[m][r] = my_xch[(32 * m + rowmap(r, hi)) * TRL_TLD + j];
    f32x16 h2 = act_tile<ACT>(layer_tile<NT, LP::LD2>(bias_tile(sp + LP::B2 + 32 * mo, hi), sp + LP::W2, mo, hh, i, hi));
    // partial head over this wave's 32 features
    const float* w3 = net == 0 ? sp_pf + LP::W3 : sp_vf + LV::W3;
    const int n_out = net == 0 ? A : 1;
#pragma unroll
    for (int o = 0; o < A; ++o) {
      if (o < n_out) {
        float p = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(w3 + o * H + 32 * mo + 8 * q + 4 * hi);
          p = fmaf(w[0], h2[4 * q + 0], p); p = fmaf(w[1], h2[4 * q + 1], p);
          p = fmaf(w[2], h2[4 * q + 2], p); p = fmaf(w[3], h2[4 * q + 3], p);
        }
        p += __shfl_xor(p, 32, 64);
        if (hi == 0) headp[((net * 2 + mo) * 8 + o) * 32 + j] = p;
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < A; ++o)
      mean[o] = headp[((0 * 2 + 0) * 8 + o) * 32 + j] + headp[((0 * 2 + 1) * 8 + o) * 32 + j] + sp_pf[LP::B3 + o];
    value = headp[((1 * 2 + 0) * 8 + 0) * 32 + j] + headp[((1 * 2 + 1) * 8 + 0) * 32 + j] + sp_vf[LV::B3];
  };

  for (int t = 0; t < a.n_steps; ++t) {
    const int row = (a.top + t) % a.rows;
    const size_t cell = (size_t)row * a.N + n;

    // exploration noise: host stream (reference parity, distribution.py:67-70) or device Philox
    float eps[A];
    if (a.deterministic) {
#pragma unroll
      for (int o = 0; o < A; ++o) eps[o] = 0.0f;
    } else if (a.noise) {
#pragma unroll
      for (int o = 0; o < A; ++o) eps[o] = valid ? a.noise[((size_t)t * a.N + n) * A + o] : 0.0f;
    } else {
      // lane half hi draws Philox block hi (4 normals); the halves swap through one cross-lane
      // exchange, so each lane pays for one block instead of ceil(A/4)
      static_assert(A <= 8, "noise split assumes at most two Philox blocks per env-step");
      const int64_t gs = a.noise_step0 + t;
      float z[4], zx[4];
      philox_normals4((uint32_t)(gs & 0xFFFFFFFFll), (uint32_t)((gs >> 32) & 0xFFFFFFFFll), (uint32_t)hi,
                      TRL_TAG_NOISE, env_seed, z);
#pragma unroll
      for (int c = 0; c < 4; ++c) zx[c] = __shfl_xor(z[c], 32, 64);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < A) eps[c] = hi ? zx[c] : z[c];
        if (4 + c < A) eps[4 + c] = hi ? z[c] : zx[c];
      }
    }

    float mean[A], value;
    forward(xb, mean, value);

    // action + log-prob under the collecting policy (continuous_policy.py:123-129, distribution.py:33-45)
    float act[A], logp = 0.0f, act_sq = 0.0f;
#pragma unroll
    for (int o = 0; o < A; ++o) {
      const float z = fmaf(stdv[o], eps[o], mean[o]);
      act[o] = a.tanh_action ? trl_tanh(z) : z;
      if (wave == 2) {                                       // only the wave that stores old_logp needs it
        float zc;
        logp += gauss_logp_term(act[o], mean[o], inv_var[o], lsv[o], a.tanh_action, zc);
      }
      act_sq = fmaf(act[o], act[o], act_sq);
    }

    // ---- env step: next^T[f][j] = sum_k M[f][k] [obs; act]^T[k][j] ----
    f32x16 acc = zero_tile();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = rowmap(s, hi);
      acc = mfma32((k < D) ? senv[i * ELD + k] : 0.0f, xb[s], acc);
    }
#pragma unroll
    for (int s = 0; s < KA; ++s) {
      const int k = 2 * s + hi;
      const float a_odd = (2 * s + 1 < A) ? act[(2 * s + 1 < A) ? 2 * s + 1 : 0] : 0.0f;
      const float av = hi ? a_odd : act[2 * s];
      acc = mfma32((k < A) ? senv[i * ELD + D + k] : 0.0f, av, acc);
    }
    float nx[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) nx[s] = (rowmap(s, hi) < D) ? trl_tanh(acc[s]) : 0.0f;

    const float nx0 = __shfl(nx[0], j, 64);                 // feature 0 lives in the hi == 0 lane
    const float raw_rew = a.reward_scale * (nx0 - 0.1f * act_sq);
    t_env += 1; cur_step += 1;
    const bool done = t_env >= a.horizon;
    const bool surpass = cur_step >= a.max_episode_frames;
    ep_ret += raw_rew;
    if (wave == 0 && hi == 0 && valid) {
      rew_sum += (double)raw_rew;
      if (done) {                                           // on_policy.py:128-130
        const int slot = atomicAdd(a.ep_count, 1);
        if (slot < a.ep_cap) { a.ep_log[slot * 3 + 0] = (float)(a.step0 + t); a.ep_log[slot * 3 + 1] = (float)n; a.ep_log[slot * 3 + 2] = ep_ret; }
      }
    }
    if (done) ep_ret = 0.0f;

    const bool flag = valid && (done || surpass);
    float st_rew = raw_rew, st_term = done ? 1.0f : 0.0f;
    float xn[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xn[s] = nx[s];
    if (__ballot(flag) != 0ull) {                           // uniform across the workgroup (state is replicated)
      float m2[A], v2;
      forward(nx, m2, v2);                                  // on_policy.py:135-143
      st_rew = raw_rew + a.discount * v2 * (surpass ? 1.0f : 0.0f);
      st_term = (done || surpass) ? 1.0f : 0.0f;
      if (flag) {                                           // partial_reset (vecenv.py:47-51) + counters (:148)
        ep_idx += 1; t_env = 0; cur_step = 0;
#pragma unroll
        for (int g = 0; g < (KS + 3) / 4; ++g) {
          float z[4];
          philox_normals4((uint32_t)ep_idx, 0u, 2 * g + hi, TRL_TAG_RESET, env_seed, z);
#pragma unroll
          for (int c = 0; c < 4; ++c) if (4 * g + c < KS) xn[4 * g + c] = (rowmap(4 * g + c, hi) < D) ? z[c] : 0.0f;
        }
      }
    }

    // ---- ring-buffer row `row` (replay_buffers/base.py:19-29), one key group per wave ----
    if (valid && a.store) {
      if (wave == 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) { const int k = rowmap(s, hi); if (k < D) a.obs[cell * D + k] = xb[s]; }
      } else if (wave == 1) {
#pragma unroll
        for (int s = 0; s < KS; ++s) { const int k = rowmap(s, hi); if (k < D) a.next_obs[cell * D + k] = nx[s]; }
      } else if (wave == 2) {
        if (hi == 0) {
#pragma unroll
          for (int o = 0; o < A; ++o) a.acts[cell * A + o] = act[o];
        } else {
          a.values[cell] = value;
          if (a.old_logp) a.old_logp[cell] = logp;
        }
      } else if (hi == 0) {
        a.rewards[cell] = st_rew; a.terminals[cell] = st_term; a.time_limits[cell] = done ? 1.0f : 0.0f;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) xb[s] = xn[s];
  }

  // ---- persist env / collector state ----
  if (wave == 0 && valid) {
#pragma unroll
    for (int s = 0; s < KS; ++s) { const int k = rowmap(s, hi); if (k < D) a.cur_obs[(size_t)n * D + k] = xb[s]; }
    if (hi == 0) { a.t_env[n] = t_env; a.cur_step[n] = cur_step; a.episode_idx[n] = ep_idx; a.ep_return[n] = ep_ret; }
  }
  if (wave == 0) {
    const double tot = wave_sum(rew_sum);
    if (lane == 0 && a.epoch_reward) atomicAdd(a.epoch_reward, tot);
  }
}

template <int D, int H, int A, int ACT>
static int launch_rollout(const RolloutDev& d, hipStream_t s) {
  using S = RoShape<D, H, A>;
  const size_t lds = S::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)rollout_kernel<D, H, A, ACT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { trl_set_error("rollout: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((rollout_kernel<D, H, A, ACT>), dim3(trl_ceil_div(d.N, 32)), dim3(RO_THREADS), lds, s, d);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
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
  hipStream_t s = (hipStream_t)stream;
  if (p->D == 17 && p->H == 64 && p->A == 6) {
    if (p->act == TRL_ACT_TANH) return launch_rollout<17, 64, 6, TRL_ACT_TANH>(d, s);
    if (p->act == TRL_ACT_RELU) return launch_rollout<17, 64, 6, TRL_ACT_RELU>(d, s);
  }
  trl_set_error("rollout: shape D=%d H=%d A=%d act=%d not instantiated", p->D, p->H, p->A, p->act);
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
