// K14 / K15 / K17 -- DQN TD loss, QR-DQN quantile-Huber loss (forward + output gradient), greedy /
// epsilon-greedy action selection, and the synthetic Atari-shaped frame env.
//
// Reference arithmetic: torchrl/algo/off_policy/dqn.py:53-60, qrdqn.py:39-60,
// torchrl/algo/utils.py:5-13 (quantile_regression_loss, huber),
// torchrl/policies/discrete_policies.py:40-67, 86-89.
#include "trl_common.h"
#include "trl_philox.h"

#define DQ_THREADS 256

__device__ __forceinline__ double dq_block_sum(double v, double* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < DQ_THREADS / 64; ++w) r += smem[w];
  return r;
}

// ---------------------------------------------------------------- K14
// q_s_a = Q(s)[a];  target = r + gamma (1 - d) max_a' Q'(s')[a'];  loss = mean (q_s_a - target)^2
// dq (B, A) is zero except 2 (q_s_a - target) / B at the taken action.
// sums (3 doubles): loss sum, q_s_a sum, reward sum.
// Actions come as int64 (`act`) or as the floats the replay buffer stores (`act_f`, exactly one is non-null); with `ring`
// the three sums are also filed into row (update count % slots) of a (slots, 3) ring, read back once per epoch.
struct DqRing { double* ring; const double* step; int slots; };
// The stored action of sample b as an index in [0, A): a NaN or out-of-range stored value (an uninitialised row, a host env
// that returned a float action) must not turn into an out-of-bounds read of q and write of dq inside a replayed graph --
// it is clamped (NaN -> 0).  The reference raises an IndexError there (dqn.py:54, `gather`); the host side validates the
// actions it stores.
__device__ __forceinline__ int dq_action(const int64_t* act, const float* act_f, int b, int A) {
  int at;
  if (act) { const int64_t a = act[b]; at = a < 0 ? 0 : (a >= A ? A - 1 : (int)a); }
  else { const float a = act_f[b]; at = (a >= 0.0f) ? (a < (float)A ? (int)a : A - 1) : 0; }
  return at;
}
__device__ __forceinline__ void dq_file(const DqRing& r, double a, double b, double c) {
  if (!r.ring) return;
  const int64_t u = (int64_t)r.step[0];                                // (the optimiser step of this update comes later)
  double* row = r.ring + 3 * (((u % r.slots) + r.slots) % r.slots);
  row[0] = a; row[1] = b; row[2] = c;
}
__global__ __launch_bounds__(DQ_THREADS) void dqn_td_kernel(const float* __restrict__ q, const int64_t* __restrict__ act,
                                                            const float* __restrict__ act_f,
                                                            const float* __restrict__ qn, const float* __restrict__ rew,
                                                            const float* __restrict__ term, float gamma, int B, int A,
                                                            float* __restrict__ dq, double* __restrict__ sums, DqRing ring) {
  __shared__ double smem[DQ_THREADS / 64];
  double sl = 0, sq = 0, sr = 0;
  const float inv_b = 1.0f / (float)B;
  for (int b = threadIdx.x; b < B; b += DQ_THREADS) {
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) mx = fmaxf(mx, qn[(size_t)b * A + a]);
    const int at = dq_action(act, act_f, b, A);
    const float qsa = q[(size_t)b * A + at];
    const float tgt = rew[b] + gamma * (1.0f - term[b]) * mx;
    const float e = qsa - tgt;
    for (int a = 0; a < A; ++a) dq[(size_t)b * A + a] = (a == at) ? 2.0f * e * inv_b : 0.0f;
    sl += (double)e * e; sq += (double)qsa; sr += (double)rew[b];
  }
  sl = dq_block_sum(sl, smem); sq = dq_block_sum(sq, smem); sr = dq_block_sum(sr, smem);
  if (threadIdx.x == 0) { sums[0] = sl; sums[1] = sq; sums[2] = sr; dq_file(ring, sl, sq, sr); }
}
static int dqn_td(const float* q, const int64_t* acts, const float* acts_f, const float* q_next, const float* rewards,
                  const float* terminals, float gamma, int B, int A, float* dq, double* sums, DqRing ring, void* stream) {
  TRL_REQUIRE(B > 0 && A > 0, "bad sizes");
  TRL_REQUIRE(q && (acts || acts_f) && q_next && rewards && terminals && dq && sums, "null pointer");
  hipLaunchKernelGGL(dqn_td_kernel, dim3(1), dim3(DQ_THREADS), 0, (hipStream_t)stream, q, acts, acts_f, q_next, rewards,
                     terminals, gamma, B, A, dq, sums, ring);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_dqn_td_loss_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                                   const float* rewards, const float* terminals, float gamma, int B, int A, float* dq,
                                   double* sums, double* ring, int slots, const double* update_count, void* stream) {
  TRL_REQUIRE(!acts != !acts_f, "dqn_td_loss: exactly one of acts (int64) / acts_f (float)");
  TRL_REQUIRE(!ring || (slots > 0 && update_count), "dqn_td_loss: ring without slots / counter");
  return dqn_td(q, acts, acts_f, q_next, rewards, terminals, gamma, B, A, dq, sums, DqRing{ring, update_count, slots}, stream);
}

// ---------------------------------------------------------------- K15
// theta = Q(s).view(B, A, Q)[a] (Q quantiles);  a* = argmax_a mean_i Q'(s')[a][i];
// T_i = r + gamma (1 - d) Q'(s')[a*][i];  diff[i][j] = T_i - theta_j;
// loss = mean_{b,i,j} huber(diff) * |tau_j - 1[diff < 0]|,  tau_j = (2j + 1) / 2Q.
// One workgroup per sample: T and theta staged in LDS, thread j owns theta_j's column of the
// Q x Q table (Q^2 = 40 000 Huber terms per sample at Q = 200 -- VALU-bound, 1.6 KB of input).
__global__ __launch_bounds__(DQ_THREADS) void quantile_huber_kernel(const float* __restrict__ q, const int64_t* __restrict__ act,
                                                                    const float* __restrict__ act_f,
                                                                    const float* __restrict__ qn, const float* __restrict__ rew,
                                                                    const float* __restrict__ term, float gamma, int B, int A,
                                                                    int Q, float* __restrict__ dq,
                                                                    double* __restrict__ part /* (B, 2): loss, q_s_a sums */) {
  extern __shared__ float sm[];                     // T[Q] | theta[Q] | means[A]
  __shared__ double smem[DQ_THREADS / 64];
  __shared__ int s_astar;
  float* T = sm;
  float* th = sm + Q;
  float* means = sm + 2 * Q;
  const int b = blockIdx.x;
  const float* qb = q + (size_t)b * A * Q;
  const float* nb = qn + (size_t)b * A * Q;
  // greedy next action by mean over quantiles (first maximal index, like torch.max)
  for (int a = threadIdx.x >> 6; a < A; a += DQ_THREADS / 64) {
    float s = 0.0f;
    for (int i = threadIdx.x & 63; i < Q; i += 64) s += nb[a * Q + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) means[a] = s / (float)Q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = 0;
    for (int a = 1; a < A; ++a) if (means[a] > means[best]) best = a;
    s_astar = best;
  }
  __syncthreads();
  const int at = dq_action(act, act_f, b, A), as = s_astar;
  const float r = rew[b], nd = gamma * (1.0f - term[b]);
  for (int i = threadIdx.x; i < Q; i += DQ_THREADS) { T[i] = r + nd * nb[as * Q + i]; th[i] = qb[at * Q + i]; }
  for (int e = threadIdx.x; e < A * Q; e += DQ_THREADS) dq[(size_t)b * A * Q + e] = 0.0f;
  __syncthreads();
  const float norm = 1.0f / ((float)B * (float)Q * (float)Q);
  double loss = 0.0, qsum = 0.0;
  for (int j = threadIdx.x; j < Q; j += DQ_THREADS) {
    const float tj = th[j], tau = (2.0f * j + 1.0f) / (2.0f * Q);
    float l = 0.0f, g = 0.0f;
    for (int i = 0; i < Q; ++i) {
      const float d = T[i] - tj;
      const float ad = fabsf(d);
      const float w = fabsf(tau - (d < 0.0f ? 1.0f : 0.0f));
      const bool quad = ad < 1.0f;
      l = fmaf(quad ? 0.5f * d * d : ad - 0.5f, w, l);
      g = fmaf(quad ? d : (d > 0.0f ? 1.0f : -1.0f), w, g);      // d huber / d diff  (d diff / d theta_j = -1)
    }
    dq[(size_t)b * A * Q + at * Q + j] = -g * norm;
    loss += (double)l; qsum += (double)tj;
  }
  loss = dq_block_sum(loss, smem); qsum = dq_block_sum(qsum, smem);
  if (threadIdx.x == 0) { part[b * 2 + 0] = loss; part[b * 2 + 1] = qsum; }
}
__global__ __launch_bounds__(DQ_THREADS) void quantile_fold_kernel(const double* __restrict__ part,
                                                                   const float* __restrict__ rew, int B,
                                                                   double* __restrict__ sums, DqRing ring) {
  __shared__ double smem[DQ_THREADS / 64];
  double a = 0, c = 0, r = 0;
  for (int b = threadIdx.x; b < B; b += DQ_THREADS) { a += part[b * 2]; c += part[b * 2 + 1]; r += (double)rew[b]; }
  a = dq_block_sum(a, smem); c = dq_block_sum(c, smem); r = dq_block_sum(r, smem);
  if (threadIdx.x == 0) { sums[0] = a; sums[1] = c; sums[2] = r; dq_file(ring, a, c, r); }
}
static int quantile_huber(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                          const float* rewards, const float* terminals, float gamma, int B, int A, int Q, float* dq,
                          double* workspace, double* sums, DqRing ring, void* stream);
extern "C" int trl_quantile_huber_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                                      const float* rewards, const float* terminals, float gamma, int B, int A, int Q,
                                      float* dq, double* workspace /* 2B doubles */, double* sums, double* ring, int slots,
                                      const double* update_count, void* stream) {
  TRL_REQUIRE(!acts != !acts_f, "quantile_huber: exactly one of acts (int64) / acts_f (float)");
  TRL_REQUIRE(!ring || (slots > 0 && update_count), "quantile_huber: ring without slots / counter");
  return quantile_huber(q, acts, acts_f, q_next, rewards, terminals, gamma, B, A, Q, dq, workspace, sums,
                        DqRing{ring, update_count, slots}, stream);
}
static int quantile_huber(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                          const float* rewards, const float* terminals, float gamma, int B, int A, int Q, float* dq,
                          double* workspace, double* sums, DqRing ring, void* stream) {
  TRL_REQUIRE(B > 0 && A > 0 && Q > 0 && A <= 64 && Q <= 4096, "bad sizes");
  TRL_REQUIRE(q && (acts || acts_f) && q_next && rewards && terminals && dq && workspace && sums, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(quantile_huber_kernel, dim3(B), dim3(DQ_THREADS), (2 * Q + A) * sizeof(float), s, q, acts, acts_f,
                     q_next, rewards, terminals, gamma, B, A, Q, dq, workspace);
  TRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(quantile_fold_kernel, dim3(1), dim3(DQ_THREADS), 0, s, workspace, rewards, B, sums, ring);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- K17
// action[n] = argmax_a score(n, a)  with score = Q (DQN) or mean over quantiles (QR-DQN);
// where u[n] < epsilon the action is replaced by rand_act[n]
// (discrete_policies.py:58-65: u = np.random.rand, rand_act = np.random.randint on the host in
// parity mode).  u / rand_act may be NULL (greedy).
__global__ __launch_bounds__(DQ_THREADS) void eps_greedy_kernel(const float* __restrict__ q, int N, int A, int Q,
                                                                const float* __restrict__ u, const int64_t* __restrict__ ra,
                                                                float epsilon, int64_t* __restrict__ action) {
  // Q == 1: a thread per env.  Quantile nets: a wave per env, lanes stride over the quantiles of an action
  // (coalesced reads of the A * Q row; a thread per env walked it with a stride of A * Q floats).
  int best = 0;
  float bv = -INFINITY;
  int n;
  if (Q == 1) {
    n = blockIdx.x * DQ_THREADS + threadIdx.x;
    if (n >= N) return;
    for (int a = 0; a < A; ++a) {
      const float s = q[(size_t)n * A + a];
      if (s > bv) { bv = s; best = a; }
    }
  } else {
    const int lane = threadIdx.x & 63;
    n = blockIdx.x * (DQ_THREADS / 64) + (threadIdx.x >> 6);
    if (n >= N) return;                                            // wave-uniform
    for (int a = 0; a < A; ++a) {
      const float* row = q + ((size_t)n * A + a) * Q;
      float s = 0.0f;
      for (int i = lane; i < Q; i += 64) s += row[i];
      s = wave_sum(s) / (float)Q;
      if (s > bv) { bv = s; best = a; }
    }
    if (lane != 0) return;
  }
  if (u && ra && u[n] < epsilon) best = (int)ra[n];
  action[n] = best;
}
extern "C" int trl_eps_greedy_i64(const float* q, int N, int A, int Q, const float* u, const int64_t* rand_act,
                                  float epsilon, int64_t* action, void* stream) {
  TRL_REQUIRE(N >= 0 && A > 0 && Q > 0, "bad sizes");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(q && action, "null pointer");
  hipLaunchKernelGGL(eps_greedy_kernel, dim3(trl_ceil_div(N, Q == 1 ? DQ_THREADS : DQ_THREADS / 64)), dim3(DQ_THREADS), 0, (hipStream_t)stream, q,
                     N, A, Q, u, rand_act, epsilon, action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- synthetic Atari-shaped env
// Frame stacks (N, C=4, 84, 84) uint8.  A step shifts the stack by one frame and appends a new
// pseudo-random frame: bytes [16 blk, 16 blk + 16) of the frame of env n at env-time t are the
// Philox4x32-10 block keyed (env_seed; t, 0, blk, 'FRME') (little-endian words).  A reset
// rebuilds the stack as the frames of times -(C-1) .. 0.  reward = 1 if action == (first byte of
// the new frame) % A else 0;  done = time_limit = (t >= horizon).  Pure throughput stand-in for
// ALE (SURVEY.md 8(d) cfg 5); one workgroup per env.
#define TRL_TAG_FRAME 0x46524D45u
__device__ __forceinline__ void synth_frame_fill(uint8_t* dst, int t, int64_t env_seed, int HW) {
  for (int blk = threadIdx.x; blk * 16 < HW; blk += DQ_THREADS) {
    uint32_t x[4];
    philox4x32_10((uint32_t)t, 0u, (uint32_t)blk, TRL_TAG_FRAME, (uint32_t)(env_seed & 0xFFFFFFFFll),
                  (uint32_t)((env_seed >> 32) & 0xFFFFFFFFll), x);
    *reinterpret_cast<uint4*>(dst + blk * 16) = make_uint4(x[0], x[1], x[2], x[3]);
  }
}
__global__ __launch_bounds__(DQ_THREADS) void synth_frames_kernel(uint8_t* __restrict__ frames, const int64_t* __restrict__ act,
                                                                  int32_t* __restrict__ t_env, int64_t seed_base, int horizon,
                                                                  int A, uint8_t* __restrict__ next_out,
                                                                  float* __restrict__ rew, float* __restrict__ done,
                                                                  const uint8_t* __restrict__ reset_mask, int reset_all,
                                                                  int N, int C, int HW) {
  const int n = blockIdx.x;
  uint8_t* f = frames + (size_t)n * C * HW;
  const int64_t env_seed = seed_base + n;
  const bool is_reset_call = reset_all || reset_mask;
  if (is_reset_call) {
    if (!reset_all && !reset_mask[n]) return;                  // block-uniform
    for (int c = 0; c < C; ++c) synth_frame_fill(f + (size_t)c * HW, c - (C - 1), env_seed, HW);
    if (threadIdx.x == 0) t_env[n] = 0;
    return;
  }
  const int t = t_env[n] + 1;
  // shift frame c <- frame c + 1 (a thread moves the same 16-byte slots of every frame, in order)
  for (int p = threadIdx.x * 16; p < HW; p += DQ_THREADS * 16)
    for (int c = 0; c + 1 < C; ++c)
      *reinterpret_cast<uint4*>(f + (size_t)c * HW + p) = *reinterpret_cast<const uint4*>(f + (size_t)(c + 1) * HW + p);
  synth_frame_fill(f + (size_t)(C - 1) * HW, t, env_seed, HW);
  __syncthreads();
  if (next_out)
    for (int p = threadIdx.x * 16; p < C * HW; p += DQ_THREADS * 16)
      *reinterpret_cast<uint4*>(next_out + (size_t)n * C * HW + p) = *reinterpret_cast<const uint4*>(f + p);
  if (threadIdx.x == 0) {
    t_env[n] = t;
    const int want = f[(size_t)(C - 1) * HW] % A;
    rew[n] = ((int)act[n] == want) ? 1.0f : 0.0f;
    done[n] = t >= horizon ? 1.0f : 0.0f;
  }
}
extern "C" int trl_synth_frames_step_u8(uint8_t* frames, const int64_t* acts, int32_t* t_env, int64_t env_seed_base,
                                        int horizon, int A, uint8_t* next_obs, float* rewards, float* dones, int N, int C,
                                        int HW, void* stream) {
  TRL_REQUIRE(N >= 0 && C > 0 && HW > 0 && HW % 16 == 0 && A > 0, "bad sizes (HW must be a multiple of 16)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(frames && t_env && acts && rewards && dones, "null pointer");
  hipLaunchKernelGGL(synth_frames_kernel, dim3(N), dim3(DQ_THREADS), 0, (hipStream_t)stream, frames, acts, t_env,
                     env_seed_base, horizon, A, next_obs, rewards, dones, (const uint8_t*)nullptr, 0, N, C, HW);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_synth_frames_reset_u8(uint8_t* frames, int32_t* t_env, int64_t env_seed_base, const uint8_t* mask,
                                         int N, int C, int HW, void* stream) {
  TRL_REQUIRE(N >= 0 && C > 0 && HW > 0 && HW % 16 == 0, "bad sizes (HW must be a multiple of 16)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(frames && t_env, "null pointer");
  hipLaunchKernelGGL(synth_frames_kernel, dim3(N), dim3(DQ_THREADS), 0, (hipStream_t)stream, frames,
                     (const int64_t*)nullptr, t_env, env_seed_base, 1, 1, (uint8_t*)nullptr, (float*)nullptr,
                     (float*)nullptr, mask, mask ? 0 : 1, N, C, HW);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
