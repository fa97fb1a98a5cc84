// K12 / K13 and friends -- the element-wise, reduction and target-update pieces of the twin-Q SAC
// update (torchrl/algo/off_policy/twin_sac_q.py:84-220) and of the off-policy collector
// (torchrl/collector/base.py:184-230).  The dense layers around them run on k_gemm.hip.
//
// All kernels are memory-bound streaming passes over (B, <=32) fp32 rows; scalar results
// (losses, means, Adam state of log_alpha) stay on the device.
#include <algorithm>
#include <cstdlib>
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_philox.h"

#define SAC_THREADS 256

// block-wide sum into thread 0 (double) -- deterministic tree; smem holds one double per wave of the block
#define SAC_WIDE 1024                               /* the single-workgroup reductions over a batch: 16 waves */
__device__ __forceinline__ double block_sum(double v, double* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += smem[w];
  return r;
}
__device__ __forceinline__ double block_max(double v, double* smem) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, smem[w]);
  return r;
}

// ---------------------------------------------------------------- concat [a | b] along features
__global__ __launch_bounds__(SAC_THREADS) void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, int rows, int fa, int fb) {
  const int f = fa + fb;
  for (int64_t e = (int64_t)blockIdx.x * SAC_THREADS + threadIdx.x; e < (int64_t)rows * f;
       e += (int64_t)gridDim.x * SAC_THREADS) {
    const int r = (int)(e / f), c = (int)(e - (int64_t)r * f);
    out[e] = c < fa ? a[(size_t)r * fa + c] : b[(size_t)r * fb + (c - fa)];
  }
}
extern "C" int trl_concat2_f32(const float* a, const float* b, float* out, int rows, int fa, int fb, void* stream) {
  TRL_REQUIRE(rows >= 0 && fa > 0 && fb > 0, "bad sizes");
  if (rows == 0) return TRL_OK;
  TRL_REQUIRE(a && b && out, "null pointer");
  int grid = trl_ceil_div((int64_t)rows * (fa + fb), SAC_THREADS);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(concat2_kernel, dim3(grid), dim3(SAC_THREADS), 0, (hipStream_t)stream, a, b, out, rows, fa, fb);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- reparameterised TanhNormal sample
// head (B, 2A) = [mean | log_std] (GuassianContPolicy.forward, continuous_policy.py:162-170: chunk,
// clamp log_std to [-20, 2]); z = mean + std * eps; action = tanh(z);
// log_prob = sum_a Normal(mean, std).log_prob(z) - log(1 - action^2 + 1e-6)   (distribution.py:33-45,
// with pre_tanh_value = z as explore(return_log_probs=True) passes it, continuous_policy.py:108-114).
__device__ __forceinline__ float rsample_row(const float* __restrict__ head, const float* __restrict__ eps,
                                             float* __restrict__ act, int A, int tanh_action) {
  float lp = 0.0f;                                 // head / eps / act point at row b
  for (int o = 0; o < A; ++o) {
    const float mean = head[o];
    const float ls = fminf(fmaxf(head[A + o], -20.0f), 2.0f);
    const float sd = __expf(ls), e = eps[o];
    const float z = fmaf(sd, e, mean);
    const float zc = z - mean;
    float lpo = -(zc * zc) / (2.0f * sd * sd) - ls - 0.91893853320467274f;
    float a = z;
    if (tanh_action) { a = trl_tanh(z); lpo -= __logf(fmaf(-a, a, 1.0f) + 1e-6f); }
    act[o] = a;
    lp += lpo;
  }
  return lp;
}
__global__ __launch_bounds__(SAC_THREADS) void rsample_fwd_kernel(const float* __restrict__ head,
                                                                  const float* __restrict__ eps,
                                                                  float* __restrict__ act, float* __restrict__ logp,
                                                                  int B, int A, int tanh_action) {
  const int b = blockIdx.x * SAC_THREADS + threadIdx.x;
  if (b >= B) return;
  logp[b] = rsample_row(head + (size_t)b * 2 * A, eps + (size_t)b * A, act + (size_t)b * A, A, tanh_action);
}
extern "C" int trl_tanh_gauss_rsample_fwd_f32(const float* head, const float* eps, float* act, float* logp, int B,
                                              int A, int tanh_action, void* stream) {
  TRL_REQUIRE(B >= 0 && A > 0, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(head && eps && act && logp, "null pointer");
  hipLaunchKernelGGL(rsample_fwd_kernel, dim3(trl_ceil_div(B, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, head, eps, act, logp, B, A, tanh_action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// row `row` (A values) of the (rows, A) standard-normal draw trl_philox_normal_f32 makes for (seed, ctr): element e is
// normal (e & 3) of Philox block e / 4
__device__ __forceinline__ void philox_noise_row(int64_t seed, int64_t ctr, int64_t row, int A, float* out) {
  const int64_t e0 = row * A;
  int64_t blk = -1;
  float z[4];
  for (int k = 0; k < A; ++k) {
    const int64_t e = e0 + k;
    if ((e >> 2) != blk) {
      blk = e >> 2;
      philox_normals4((uint32_t)(ctr & 0xFFFFFFFFll), (uint32_t)((ctr >> 32) & 0xFFFFFFFFll), (uint32_t)blk, TRL_TAG_NOISE,
                      seed, z);
    }
    out[k] = z[e & 3];
  }
}

// Both policy samples of one SAC update and the three critic inputs in one launch (twin_sac_q.py:93-106, :125-131,
// :146-151): new_a / logp from head(obs) with eps1, next_a / next_logp from head(next_obs) with eps2, and
// x_sa = [obs | acts], x_next = [next_obs | next_a], x_new = [obs | new_a].  One thread per batch row -- the whole
// batch is ~1 MB, what is saved is four dependent launches.
__global__ __launch_bounds__(SAC_THREADS) void sac_samples_kernel(const float* __restrict__ head, const float* __restrict__ head2,
                                                                  const float* __restrict__ eps1, const float* __restrict__ eps2,
                                                                  const float* __restrict__ obs, const float* __restrict__ acts,
                                                                  const float* __restrict__ nobs, float* __restrict__ new_a,
                                                                  float* __restrict__ logp, float* __restrict__ next_a,
                                                                  float* __restrict__ next_logp, float* __restrict__ x_sa,
                                                                  float* __restrict__ x_next, float* __restrict__ x_new,
                                                                  int B, int D, int A, int tanh_action,
                                                                  const double* __restrict__ step_state, int64_t seed,
                                                                  float* __restrict__ eps1_out,
                                                                  double* __restrict__ mom_part) {
  // two threads per batch row: thread (b, 0) draws / samples from head(obs) and writes x_sa, x_new; thread (b, 1) does the
  // next-state pair and x_next -- the Philox + Box-Muller chain per draw is what this kernel spends its time in
  const int t = blockIdx.x * blockDim.x + threadIdx.x;               // (whole waves take one side: no divergence)
  const int b = ((t >> 7) << 6) | (t & 63), which = (t >> 6) & 1;
  const bool stats = mom_part && which == 0, live = b < B;           // (a stats wave stays whole for its reductions)
  if (!live && !stats) return;
  float lp0 = 0.0f;
  if (live) {
    const int F = D + A;
    float e[8];
    if (step_state) {
      // the two draws of update u (u = optimiser steps taken so far, device-resident: the launch is graph-replayed) are
      // trl_philox_normal_f32(seed, 2 u + 1) and (seed, 2 u + 2) -- what the engine launched separately before
      const int64_t u = (int64_t)step_state[0];
      philox_noise_row(seed, 2 * u + 1 + which, b, A, e);
      if (which == 0) for (int k = 0; k < A; ++k) eps1_out[(size_t)b * A + k] = e[k];   // the sampler's backward pass reads it
    } else {
      const float* src = which == 0 ? eps1 : eps2;
      for (int k = 0; k < A; ++k) e[k] = src[(size_t)b * A + k];
    }
    if (which == 0) {
      float* na = new_a + (size_t)b * A;
      lp0 = rsample_row(head + (size_t)b * 2 * A, e, na, A, tanh_action);
      logp[b] = lp0;
      for (int k = 0; k < D; ++k) {
        const float o = obs[(size_t)b * D + k];
        x_sa[(size_t)b * F + k] = o; x_new[(size_t)b * F + k] = o;
      }
      for (int k = 0; k < A; ++k) {
        x_sa[(size_t)b * F + D + k] = acts[(size_t)b * A + k];
        x_new[(size_t)b * F + D + k] = na[k];
      }
    } else {
      float* xa = next_a + (size_t)b * A;
      next_logp[b] = rsample_row(head2 + (size_t)b * 2 * A, e, xa, A, tanh_action);
      for (int k = 0; k < D; ++k) x_next[(size_t)b * F + k] = nobs[(size_t)b * D + k];
      for (int k = 0; k < A; ++k) x_next[(size_t)b * F + D + k] = xa[k];
    }
  }
  if (stats) {
    // the logged moments of the policy head on obs -- clamped log_std, mean -- and of log_prob (twin_sac_q.py:190-207) as
    // per-wave partials {sum, sum of squares, max, -min} x {log_std, log_prob, mean}: row b >> 6 of mom_part (12 doubles);
    // trl_sac_losses_f32 folds them (the separate trl_moments_multi_f64 launch re-read head and logp for this)
    double ps[3] = {0, 0, 0}, pq[3] = {0, 0, 0}, pm[3] = {-INFINITY, -INFINITY, -INFINITY}, pn[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (live) {
      for (int k = 0; k < A; ++k) {
        const double mu = (double)head[(size_t)b * 2 * A + k];
        const double ls = (double)fminf(fmaxf(head[(size_t)b * 2 * A + A + k], -20.0f), 2.0f);
        ps[0] += ls; pq[0] += ls * ls; pm[0] = fmax(pm[0], ls); pn[0] = fmax(pn[0], -ls);
        ps[2] += mu; pq[2] += mu * mu; pm[2] = fmax(pm[2], mu); pn[2] = fmax(pn[2], -mu);
      }
      ps[1] = (double)lp0; pq[1] = (double)lp0 * lp0; pm[1] = (double)lp0; pn[1] = -(double)lp0;
    }
    double* row = mom_part + (size_t)(b >> 6) * 12;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double a0 = wave_sum(ps[j]), a1 = wave_sum(pq[j]), a2 = wave_max(pm[j]), a3 = wave_max(pn[j]);
      if ((threadIdx.x & 63) == 0) { row[4 * j] = a0; row[4 * j + 1] = a1; row[4 * j + 2] = a2; row[4 * j + 3] = a3; }
    }
  }
}
static int sac_samples_impl(const float* head, const float* head2, const float* eps1, const float* eps2,
                            const float* obs, const float* acts, const float* next_obs, float* new_a, float* logp,
                            float* next_a, float* next_logp, float* x_sa, float* x_next, float* x_new, int B, int D,
                            int A, int tanh_action, const double* step_state, int64_t seed, float* eps1_out,
                            double* mom_part, void* stream) {
  TRL_REQUIRE(B >= 0 && A > 0 && A <= 8 && D > 0, "bad sizes (A <= 8)");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(head && head2 && obs && acts && next_obs && ((eps1 && eps2) || (step_state && eps1_out)), "null input");
  TRL_REQUIRE(new_a && logp && next_a && next_logp && x_sa && x_next && x_new, "null output");
  hipLaunchKernelGGL(sac_samples_kernel, dim3(2 * trl_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, head, head2, eps1,
                     eps2, obs, acts, next_obs, new_a, logp, next_a, next_logp, x_sa, x_next, x_new, B, D, A, tanh_action,
                     step_state, seed, eps1_out, mom_part);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
// Both policy samples of one update and the three critic inputs in ONE launch.  step_state NULL: eps1 / eps2 are read;
// step_state given: update u (u = step_state[0], the device-resident count of optimiser steps taken) draws
// trl_philox_normal_f32's values for (seed, 2 u + 1) and (seed, 2 u + 2) in place and eps1 (B, A) receives the first draw
// for the sampler's backward pass.  mom_part (nullable): the logged moments' per-wave partials as a by-product,
// ceil(B / 64) rows of 12 doubles (folded by trl_sac_losses_f32).
extern "C" int trl_sac_samples_f32(const float* head, const float* head2, float* eps1, const float* eps2,
                                   const double* step_state, int64_t seed, const float* obs, const float* acts,
                                   const float* next_obs, float* new_a, float* logp, float* next_a, float* next_logp,
                                   float* x_sa, float* x_next, float* x_new, int B, int D, int A, int tanh_action,
                                   double* mom_part, void* stream) {
  if (step_state)
    return sac_samples_impl(head, head2, nullptr, nullptr, obs, acts, next_obs, new_a, logp, next_a, next_logp, x_sa, x_next,
                            x_new, B, D, A, tanh_action, step_state, seed, eps1, mom_part, stream);
  return sac_samples_impl(head, head2, eps1, eps2, obs, acts, next_obs, new_a, logp, next_a, next_logp, x_sa, x_next, x_new,
                          B, D, A, tanh_action, nullptr, 0, nullptr, mom_part, stream);
}

// backward of the above + the std / mean regularisers of twin_sac_q.py:157-160:
//   L += w_std * mean(log_std^2) + w_mean * mean(mean^2)
// d_act (B, A) and d_logp (scalar, the same for every row: alpha / B) come from the policy loss.
// With z = mean + std eps held through eps: d logN/d mean = 0, d logN/d log_std = -1; the tanh
// correction contributes t = 2 a (1 - a^2) / (1 - a^2 + 1e-6) through z.
__global__ __launch_bounds__(SAC_THREADS) void rsample_bwd_kernel(const float* __restrict__ head,
                                                                  const float* __restrict__ eps,
                                                                  const float* __restrict__ act,
                                                                  const float* __restrict__ d_act,
                                                                  const float* __restrict__ d_act2, int ld, int off,
                                                                  const float* __restrict__ d_logp_ptr, float d_logp_mul,
                                                                  float w_std, float w_mean,
                                                                  float* __restrict__ d_head, int B, int A,
                                                                  int tanh_action) {
  const int b = blockIdx.x * SAC_THREADS + threadIdx.x;
  if (b >= B) return;
  const float d_logp = (d_logp_ptr ? *d_logp_ptr : 1.0f) * d_logp_mul;     // alpha / B, alpha a device scalar
  const float reg = 2.0f / ((float)B * (float)A);
  for (int o = 0; o < A; ++o) {
    const float mean = head[(size_t)b * 2 * A + o];
    const float raw = head[(size_t)b * 2 * A + A + o];
    const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
    const float pass = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
    const float se = __expf(ls) * eps[(size_t)b * A + o];
    const float a = act[(size_t)b * A + o];
    float da_dz = 1.0f, t = 0.0f;
    if (tanh_action) { da_dz = fmaf(-a, a, 1.0f); t = 2.0f * a * da_dz / (da_dz + 1e-6f); }
    // d(loss)/d(action) = columns [off, off + A) of d_act (+ d_act2: the twin critics' input gradients, :152-155)
    float da = d_act[(size_t)b * ld + off + o];
    if (d_act2) da += d_act2[(size_t)b * ld + off + o];
    const float g_z = da * da_dz + d_logp * t;                         // through z
    d_head[(size_t)b * 2 * A + o] = g_z + w_mean * reg * mean;
    d_head[(size_t)b * 2 * A + A + o] = pass * (g_z * se - d_logp + w_std * reg * ls);
  }
}
extern "C" int trl_tanh_gauss_rsample_bwd_f32(const float* head, const float* eps, const float* act, const float* d_act,
                                              const float* d_logp_ptr, float d_logp_mul, float w_std, float w_mean,
                                              float* d_head, int B, int A, int tanh_action, void* stream) {
  TRL_REQUIRE(B >= 0 && A > 0, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(head && eps && act && d_act && d_head, "null pointer");
  hipLaunchKernelGGL(rsample_bwd_kernel, dim3(trl_ceil_div(B, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, head, eps, act, d_act, (const float*)nullptr, A, 0, d_logp_ptr, d_logp_mul, w_std,
                     w_mean, d_head, B, A, tanh_action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
// the same with d_act = dx1[:, off:off+A] + dx2[:, off:off+A] read in place (rows of ld floats; dx2 nullable): the input
// gradients of the twin critics on [obs | new_a] need no slice-and-add launch in between
extern "C" int trl_tanh_gauss_rsample_bwd_cols_f32(const float* head, const float* eps, const float* act, const float* dx1,
                                                   const float* dx2, int ld, int off, const float* d_logp_ptr,
                                                   float d_logp_mul, float w_std, float w_mean, float* d_head, int B, int A,
                                                   int tanh_action, void* stream) {
  TRL_REQUIRE(B >= 0 && A > 0 && off >= 0 && off + A <= ld, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(head && eps && act && dx1 && d_head, "null pointer");
  hipLaunchKernelGGL(rsample_bwd_kernel, dim3(trl_ceil_div(B, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, head, eps, act, dx1, dx2, ld, off, d_logp_ptr, d_logp_mul, w_std, w_mean, d_head, B,
                     A, tanh_action);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// The policy gradient's way from the critics' first hidden layer down to the policy head in ONE streaming launch
// (twin_sac_q.py:146-160): with dZ_i the gradient at critic i's first layer output (pre-activation side: dY_i * act'(Y_i),
// Y_i the stored activations) and W_i (H, D + A) that layer's weight,
//     d_act = sum_i dZ_i W_i[:, off : off + A]          -- the ACTION columns of the input gradient; nobody reads the rest
//     d_head = rsample_bwd(d_act, ...)                  -- as rsample_bwd_kernel above
// This was a (B x H) . (H x (D + A)) GEMM per critic on 64 workgroups -- 15 us of latency for 6 useful output columns of
// 23 -- and the sampler's backward launch behind it.  Here a wave owns a row: 4 x H floats stream through, the two
// A x H weight slices sit in LDS (transposed), the A dot products are reduced by one butterfly, lanes 0..A-1 finish.
#define PG_MAX_A 8
#define PG_MAX_CH 4                                 // H <= 1024
struct PolGrad {
  const float* dy[2]; const float* y[2]; const float* w[2];   // critic i: dY, Y (B, H); W (H, ldw); y[i] NULL: dy is dZ already
  int n, H, ldw, off, gate_act;
  // optional: the policy's own head backward rides along -- dZ2 = (d_head W3) * act'(H2), the gradient at the policy's second
  // hidden layer ALREADY gated for the layer below: hw3 (2A, H) the head's weight, hh2 (B, H) that layer's outputs, hdz (B, H)
  const float* hw3; const float* hh2; float* hdz; int hact;
};
#ifdef TRL_EXP_CLK                                  // development aid (tools/bench_policy_grad.py): 100 MHz stamps of 2 workgroups
__device__ long long g_pg_clk[2 * 8];
#define PCLK(ph) if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_pg_clk[(blockIdx.x ? 8 : 0) + (ph)] = wall_clock64();
extern "C" int trl_dbg_pg_clk(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pg_clk), sizeof(long long) * 16); }
#else
#define PCLK(ph)
#endif
template <int CH>                                   // chunks of 256 floats per row: H <= 256 CH
__global__ __launch_bounds__(SAC_THREADS) void sac_policy_grad_kernel(PolGrad g, const float* __restrict__ head,
                                                                      const float* __restrict__ eps, const float* __restrict__ act,
                                                                      const float* __restrict__ d_logp_ptr, float d_logp_mul,
                                                                      float w_std, float w_mean, float* __restrict__ d_head,
                                                                      int B, int A, int tanh_action) {
  extern __shared__ __attribute__((aligned(16))) float wt[];          // [critic][PG_MAX_A][H] | head weight [2A][H] (g.hdz)
  const int H = g.H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const w3s = wt + g.n * PG_MAX_A * H;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  // a row's loads: 2 critics x {dY, Y} x 16 bytes per lane and chunk, and the sampler's inputs for lanes 0..A-1 -- requested
  // for the FIRST row before the weights are staged (nothing of a row depends on them), and for the next row before the
  // current one is reduced: one memory round trip per row instead of three
  f32x4 zv[2][CH], yv[2][CH], hv[CH];
  float h_mean = 0.0f, h_raw = 0.0f, h_eps = 0.0f, h_act = 0.0f;
  auto fetch = [&](int b) {
    if (g.hdz) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 256 + lane * 4;
        hv[c] = j < H ? *reinterpret_cast<const f32x4*>(g.hh2 + (size_t)b * H + j) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 256 + lane * 4;
        const bool on = i < g.n && j < H;
        zv[i][c] = on ? *reinterpret_cast<const f32x4*>(g.dy[i] + (size_t)b * H + j) : zero4;
        yv[i][c] = (on && g.y[i]) ? *reinterpret_cast<const f32x4*>(g.y[i] + (size_t)b * H + j) : zero4;
      }
    if (lane < A) {
      h_mean = head[(size_t)b * 2 * A + lane]; h_raw = head[(size_t)b * 2 * A + A + lane];
      h_eps = eps[(size_t)b * A + lane]; h_act = act[(size_t)b * A + lane];
    }
  };
  const int stride = gridDim.x * (SAC_THREADS / 64);
  int b = blockIdx.x * (SAC_THREADS / 64) + wave;
  PCLK(0)
  if (b < B) fetch(b);
  PCLK(1)
  // (critic index static: a per-lane index into the argument struct's pointer array costs a dependent load per element;
  //  o fastest: a weight row's A columns are one run; slices padded to PG_MAX_A rows of zeros: no `o < A` in the row loop)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < g.n) {
      const float* __restrict__ wsrc = g.w[i] + g.off;
#pragma unroll 4
      for (int r = threadIdx.x; r < A * H; r += SAC_THREADS) {
        const int j = r / A, o = r - j * A;
        wt[(i * PG_MAX_A + o) * H + j] = wsrc[(size_t)j * g.ldw + o];
      }
      for (int r = A * H + threadIdx.x; r < PG_MAX_A * H; r += SAC_THREADS) wt[i * PG_MAX_A * H + r] = 0.0f;
    }
  }
  if (g.hdz) for (int r = threadIdx.x; r < 2 * A * (H >> 2); r += SAC_THREADS)
    reinterpret_cast<f32x4*>(w3s)[r] = reinterpret_cast<const f32x4*>(g.hw3)[r];
  __syncthreads();
  PCLK(2)
  const float d_logp = (d_logp_ptr ? *d_logp_ptr : 1.0f) * d_logp_mul;     // alpha / B, alpha a device scalar
  const float reg = 2.0f / ((float)B * (float)A);
  for (; b < B; b += stride) {
    float part[PG_MAX_A];
#pragma unroll
    for (int o = 0; o < PG_MAX_A; ++o) part[o] = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < g.n) {
        if (g.y[i]) {                                // act'(Y) from the stored outputs (uniform branches, once per row)
          if (g.gate_act == TRL_ACT_RELU) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) zv[i][c][q] = yv[i][c][q] > 0.0f ? zv[i][c][q] : 0.0f;
          } else if (g.gate_act == TRL_ACT_TANH) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) zv[i][c][q] *= 1.0f - yv[i][c][q] * yv[i][c][q];
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int j = c * 256 + lane * 4;
          if (CH == 1 || j < H) {                    // (out-of-range lanes hold zeros: only the LDS address must stay inside)
            const f32x4 z = zv[i][c];
            const int j4 = (j < H ? j : 0) >> 2;
#pragma unroll
            for (int o = 0; o < PG_MAX_A; ++o) {
              const f32x4 wv = reinterpret_cast<const f32x4*>(wt)[(i * PG_MAX_A + o) * (H >> 2) + j4];
              part[o] = fmaf(z[0], wv[0], part[o]); part[o] = fmaf(z[1], wv[1], part[o]);
              part[o] = fmaf(z[2], wv[2], part[o]); part[o] = fmaf(z[3], wv[3], part[o]);
            }
          }
        }
      }
    }
    PCLK(3)
    const float mean = h_mean, raw = h_raw, ev = h_eps, a = h_act;
    f32x4 hcur[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) hcur[c] = hv[c];
    if (b + stride < B) fetch(b + stride);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
      for (int o = 0; o < PG_MAX_A; ++o) part[o] += __shfl_xor(part[o], s, 64);
    }
    float da = 0.0f, g_mean = 0.0f, g_ls = 0.0f;
#pragma unroll
    for (int o = 0; o < PG_MAX_A; ++o) da = lane == o ? part[o] : da;
    if (lane < A) {
      const int o = lane;
      const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
      const float pass = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
      const float se = __expf(ls) * ev;
      float da_dz = 1.0f, t = 0.0f;
      if (tanh_action) { da_dz = fmaf(-a, a, 1.0f); t = 2.0f * a * da_dz / (da_dz + 1e-6f); }
      const float g_z = da * da_dz + d_logp * t;                         // through z
      g_mean = g_z + w_mean * reg * mean;
      g_ls = pass * (g_z * se - d_logp + w_std * reg * ls);
      d_head[(size_t)b * 2 * A + o] = g_mean;
      d_head[(size_t)b * 2 * A + A + o] = g_ls;
    }
    if (g.hdz) {
      // dZ2[b][j] = (sum_k d_head[b][k] W3[k][j]) * act'(H2[b][j]), k ascending (means, then log_stds): lane owns 4 columns
      // per chunk, the 2A head gradients are wave-uniform (read from the lanes that made them)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 256 + lane * 4;
        if (CH == 1 || j < H) {
          const int j4 = (j < H ? j : 0) >> 2;
          f32x4 acc = zero4;
          for (int o = 0; o < A; ++o) {
            const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_mean), o));
            const f32x4 wv = reinterpret_cast<const f32x4*>(w3s)[o * (H >> 2) + j4];
            acc[0] = fmaf(m, wv[0], acc[0]); acc[1] = fmaf(m, wv[1], acc[1]); acc[2] = fmaf(m, wv[2], acc[2]); acc[3] = fmaf(m, wv[3], acc[3]);
          }
          for (int o = 0; o < A; ++o) {
            const float l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_ls), o));
            const f32x4 wv = reinterpret_cast<const f32x4*>(w3s)[(A + o) * (H >> 2) + j4];
            acc[0] = fmaf(l, wv[0], acc[0]); acc[1] = fmaf(l, wv[1], acc[1]); acc[2] = fmaf(l, wv[2], acc[2]); acc[3] = fmaf(l, wv[3], acc[3]);
          }
          const f32x4 h = hcur[c];
          if (g.hact == TRL_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = h[q] > 0.0f ? acc[q] : 0.0f;
          } else if (g.hact == TRL_ACT_TANH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= 1.0f - h[q] * h[q];
          }
          if (j < H) *reinterpret_cast<f32x4*>(g.hdz + (size_t)b * H + j) = acc;
        }
      }
    }
    PCLK(4)
  }
}
extern "C" int trl_sac_policy_grad_supported(int H, int A) {
  return H >= 4 && (H & 3) == 0 && H <= 256 * PG_MAX_CH && A >= 1 && A <= PG_MAX_A;
}
extern "C" int trl_sac_policy_grad_f32(int n, const float* const* dy, const float* const* y, int gate_act,
                                       const float* const* w, int H, int ldw, int off, const float* head, const float* eps,
                                       const float* act, const float* d_logp_ptr, float d_logp_mul, float w_std, float w_mean,
                                       float* d_head, int B, int A, int tanh_action, const float* head_w, const float* head_h,
                                       int head_act, float* head_dz, void* stream) {
  TRL_REQUIRE(B >= 0 && (n == 1 || n == 2) && trl_sac_policy_grad_supported(H, A), "policy_grad: 1-2 critics, H % 4 == 0, H <= 1024, A <= 8");
  TRL_REQUIRE(off >= 0 && off + A <= ldw, "policy_grad: action columns outside the weight");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(dy && w && head && eps && act && d_head, "null pointer");
  TRL_REQUIRE(gate_act == TRL_ACT_TANH || gate_act == TRL_ACT_RELU || gate_act == TRL_ACT_NONE, "unknown activation");
  PolGrad g{};
  g.n = n; g.H = H; g.ldw = ldw; g.off = off; g.gate_act = gate_act;
  for (int i = 0; i < n; ++i) {
    TRL_REQUIRE(dy[i] && w[i], "null pointer");
    g.dy[i] = dy[i]; g.y[i] = (y && gate_act != TRL_ACT_NONE) ? y[i] : nullptr; g.w[i] = w[i];
    TRL_REQUIRE(((reinterpret_cast<uintptr_t>(g.dy[i]) | reinterpret_cast<uintptr_t>(g.y[i])) & 15) == 0, "policy_grad: 16-byte aligned rows");
  }
  if (head_dz) {
    TRL_REQUIRE(head_w && head_h, "policy_grad: head backward needs the head weight and the hidden outputs");
    TRL_REQUIRE(head_act == TRL_ACT_TANH || head_act == TRL_ACT_RELU || head_act == TRL_ACT_NONE, "unknown activation");
    TRL_REQUIRE(((reinterpret_cast<uintptr_t>(head_w) | reinterpret_cast<uintptr_t>(head_h) | reinterpret_cast<uintptr_t>(head_dz)) & 15) == 0,
                "policy_grad: 16-byte aligned head operands");
    g.hw3 = head_w; g.hh2 = head_h; g.hdz = head_dz; g.hact = head_act;
  }
  const int lds = (n * PG_MAX_A + (head_dz ? 2 * A : 0)) * H * (int)sizeof(float);
  // (a row is one memory round trip of its wave: as many waves as the chip holds, one or two rows each)
  const int grid = std::max(1, std::min(trl_ceil_div(B, SAC_THREADS / 64), 1024));
  if (H <= 256) {
    hipLaunchKernelGGL(sac_policy_grad_kernel<1>, dim3(grid), dim3(SAC_THREADS), lds, (hipStream_t)stream, g, head, eps, act,
                       d_logp_ptr, d_logp_mul, w_std, w_mean, d_head, B, A, tanh_action);
  } else {
    static int attr_lds = 0;
    if (lds > attr_lds && lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)sac_policy_grad_kernel<PG_MAX_CH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) { trl_set_error("policy_grad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_lds = lds;
    }
    hipLaunchKernelGGL(sac_policy_grad_kernel<PG_MAX_CH>, dim3(std::min(grid, 512)), dim3(SAC_THREADS), lds, (hipStream_t)stream,
                       g, head, eps, act, d_logp_ptr, d_logp_mul, w_std, w_mean, d_head, B, A, tanh_action);
  }
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- entropy temperature step
// alpha_loss = -mean(log_alpha * (log_prob + target_entropy)); Adam step on log_alpha;
// alpha = exp(log_alpha) AFTER the step (twin_sac_q.py:111-120, its Q19 ordering).
// state: [log_alpha, exp_avg, exp_avg_sq, step]; out: [alpha, alpha_loss].   Single workgroup of 16 waves: a thread's loads
// are dependent round trips (~0.5 us each), so the batch is spread over as many threads as a workgroup has.
__global__ __launch_bounds__(SAC_WIDE) void sac_alpha_kernel(const float* __restrict__ logp, int B,
                                                                float target_entropy, float lr, float beta1,
                                                                float beta2, float eps, float* __restrict__ state,
                                                                float* __restrict__ out) {
  __shared__ double smem[SAC_WIDE / 64];
  double s = 0.0;
  for (int b0 = threadIdx.x; b0 < B; b0 += 4 * SAC_WIDE) {             // four loads in flight, summed in ascending b
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = b0 + j * SAC_WIDE < B ? logp[b0 + j * SAC_WIDE] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (b0 + j * SAC_WIDE < B) s += (double)v[j];
  }
  s = block_sum(s, smem);
  if (threadIdx.x == 0) {
    const float mean_term = (float)(s / B) + target_entropy;     // mean(log_prob + H_target)
    const float la = state[0];
    out[1] = -la * mean_term;                                    // alpha_loss
    const float g = -mean_term;
    const float t = state[3] + 1.0f;
    const float m = beta1 * state[1] + (1.0f - beta1) * g;
    const float v = beta2 * state[2] + (1.0f - beta2) * g * g;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float la_new = la - (lr / bc1) * (m / (sqrtf(v) / sqrtf(bc2) + eps));
    state[0] = la_new; state[1] = m; state[2] = v; state[3] = t;
    out[0] = __expf(la_new);
  }
}
extern "C" int trl_sac_alpha_step_f32(const float* logp, int B, float target_entropy, float lr, float beta1,
                                      float beta2, float eps, float* state, float* out, void* stream) {
  TRL_REQUIRE(B > 0, "empty batch");
  TRL_REQUIRE(logp && state && out, "null pointer");
  hipLaunchKernelGGL(sac_alpha_kernel, dim3(1), dim3(SAC_WIDE), 0, (hipStream_t)stream, logp, B, target_entropy,
                     lr, beta1, beta2, eps, state, out);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- TD target + twin MSE + policy-loss gradients
// q_target = r + (1 - d) gamma (min(tq1, tq2) - alpha logp')          (twin_sac_q.py:125-139)
// qf_i loss = mean((q_i - q_target)^2), dq_i = 2 (q_i - q_target) / B   (nn.MSELoss, :142-143)
// policy_loss = mean(alpha logp - min(q1n, q2n)); torch.min ties split the gradient (:145-155)
//   dq1n = -(q1n < q2n ? 1 : q1n == q2n ? .5 : 0) / B,  dq2n likewise.
// alpha_ptr: device scalar written by the alpha step (or a constant 1 when tuning is off).
// sums (double[4]): qf1 loss sum, qf2 loss sum, sum(alpha logp - min q_new), sum rewards.
// Extras of trl_sac_losses_f32, each optional:
//  * the temperature step (sac_alpha_kernel's arithmetic) at the top, its alpha used below -- one launch less;
//  * the fold of trl_sac_samples_f32's per-wave partial moments into {mean, unbiased std, max, min} x {log_std,
//    log_prob, mean} (12 doubles at mom_out), done by the last wave while the others reduce the loss sums.
struct LossExtras {
  float* alpha_state; float* alpha_out; float target_entropy, lr, beta1, beta2, eps;       // alpha_state NULL: no step
  const double* mom_part; int parts; int A; double* mom_out;                               // mom_part NULL: no fold
};
__device__ __forceinline__ float alpha_adam_step(double logp_sum, int B, const LossExtras& x) {
  const float mean_term = (float)(logp_sum / B) + x.target_entropy;     // mean(log_prob + H_target)
  float* state = x.alpha_state;
  const float la = state[0];
  x.alpha_out[1] = -la * mean_term;                                     // alpha_loss
  const float g = -mean_term;
  const float t = state[3] + 1.0f;
  const float m = x.beta1 * state[1] + (1.0f - x.beta1) * g;
  const float v = x.beta2 * state[2] + (1.0f - x.beta2) * g * g;
  const float bc1 = 1.0f - powf(x.beta1, t), bc2 = 1.0f - powf(x.beta2, t);
  const float la_new = la - (x.lr / bc1) * (m / (sqrtf(v) / sqrtf(bc2) + x.eps));
  state[0] = la_new; state[1] = m; state[2] = v; state[3] = t;
  const float alpha = __expf(la_new);
  x.alpha_out[0] = alpha;
  return alpha;
}
__global__ __launch_bounds__(SAC_WIDE) void sac_losses_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                                 const float* __restrict__ tq1, const float* __restrict__ tq2,
                                                                 const float* __restrict__ logp_next,
                                                                 const float* __restrict__ rew, const float* __restrict__ term,
                                                                 const float* __restrict__ q1n, const float* __restrict__ q2n,
                                                                 const float* __restrict__ logp, const float* __restrict__ alpha_ptr,
                                                                 float gamma, int B, float* __restrict__ dq1,
                                                                 float* __restrict__ dq2, float* __restrict__ dq1n,
                                                                 float* __restrict__ dq2n, double* __restrict__ sums,
                                                                 LossExtras x) {
  __shared__ double smem[SAC_WIDE / 64];
  __shared__ float s_alpha;
  // Rows are taken four per thread (b, b + 1024, ...): all forty loads of a group are in flight together, and the first
  // group is requested BEFORE the temperature step, whose reduction and one-thread Adam arithmetic then run under those
  // loads instead of in front of them (a thread's accumulation order -- ascending b -- is that of the row-by-row loop).
  constexpr int R = 4;
  float in[R][10];
  auto request = [&](int b0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int b = b0 + j * SAC_WIDE;
      if (b < B) {
        in[j][0] = tq1[b]; in[j][1] = tq2[b]; in[j][2] = logp_next[b]; in[j][3] = rew[b]; in[j][4] = term[b];
        in[j][5] = q1[b]; in[j][6] = q2[b]; in[j][7] = q1n[b]; in[j][8] = q2n[b]; in[j][9] = logp[b];
      }
    }
  };
  request(threadIdx.x);
  // the moment fold's first 64 partial rows (one per lane of the last wave) travel with the first group of loads
  double mp[3][4] = {{0, 0, -INFINITY, -INFINITY}, {0, 0, -INFINITY, -INFINITY}, {0, 0, -INFINITY, -INFINITY}};
  if (x.mom_part && (threadIdx.x >> 6) == (blockDim.x >> 6) - 1 && (int)(threadIdx.x & 63) < x.parts) {
    const double* row = x.mom_part + (size_t)(threadIdx.x & 63) * 12;
#pragma unroll
    for (int j = 0; j < 3; ++j) { mp[j][0] = row[4 * j]; mp[j][1] = row[4 * j + 1]; mp[j][2] = row[4 * j + 2]; mp[j][3] = row[4 * j + 3]; }
  }
  float alpha;
  if (x.alpha_state) {                                                 // (one workgroup: the launcher's contract)
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) if ((int)threadIdx.x + j * SAC_WIDE < B) s += (double)in[j][9];
    for (int b = threadIdx.x + R * SAC_WIDE; b < B; b += SAC_WIDE) s += (double)logp[b];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) s_alpha = alpha_adam_step(s, B, x);
    __syncthreads();
    alpha = s_alpha;
  } else {
    alpha = *alpha_ptr;
  }
  const float inv_b = 1.0f / (float)B;
  double s1 = 0, s2 = 0, sp = 0, sr = 0;
  for (int b0 = threadIdx.x; b0 < B; b0 += R * SAC_WIDE) {
    if (b0 != (int)threadIdx.x) request(b0);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int b = b0 + j * SAC_WIDE;
      if (b >= B) break;
      const float tv = fminf(in[j][0], in[j][1]) - alpha * in[j][2];
      const float qt = in[j][3] + (1.0f - in[j][4]) * gamma * tv;
      const float e1 = in[j][5] - qt, e2 = in[j][6] - qt;
      dq1[b] = 2.0f * e1 * inv_b; dq2[b] = 2.0f * e2 * inv_b;
      const float a = in[j][7], c = in[j][8];
      dq1n[b] = -(a < c ? 1.0f : (a == c ? 0.5f : 0.0f)) * inv_b;
      dq2n[b] = -(c < a ? 1.0f : (a == c ? 0.5f : 0.0f)) * inv_b;
      s1 += (double)e1 * e1; s2 += (double)e2 * e2; sp += (double)(alpha * in[j][9] - fminf(a, c)); sr += (double)in[j][3];
    }
  }
  // the four loss sums: ONE exchange (wave sums side by side, one barrier, threads 0..3 add the waves in block_sum's order)
  // instead of four block_sum calls with two barriers each
  __shared__ double red4[SAC_WIDE / 64][4];
  {
    const double w1 = wave_sum(s1), w2 = wave_sum(s2), wp = wave_sum(sp), wr = wave_sum(sr);
    if ((threadIdx.x & 63) == 0) { double* r = red4[threadIdx.x >> 6]; r[0] = w1; r[1] = w2; r[2] = wp; r[3] = wr; }
  }
  __syncthreads();
  if (threadIdx.x < 4) {                                               // ONE workgroup (launcher)
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red4[w][threadIdx.x];
    sums[threadIdx.x] = r;
  }
  if (x.mom_part && (threadIdx.x >> 6) == (blockDim.x >> 6) - 1) {     // the last wave folds the partial moments (values it
    const int lane = threadIdx.x & 63;                                 // requested at kernel entry: mp[])
    for (int j = 0; j < 3; ++j) {
      double a = mp[j][0], c = mp[j][1], mx = mp[j][2], nmn = mp[j][3];
      for (int p = lane + 64; p < x.parts; p += 64) {
        const double* row = x.mom_part + (size_t)p * 12 + 4 * j;
        a += row[0]; c += row[1]; mx = fmax(mx, row[2]); nmn = fmax(nmn, row[3]);
      }
      a = wave_sum(a); c = wave_sum(c); mx = wave_max(mx); nmn = wave_max(nmn);
      if (lane == 0) {
        const double cnt = (double)B * (j == 1 ? 1 : x.A), mean = a / cnt;
        double* o = x.mom_out + 4 * j;
        o[0] = mean;
        o[1] = cnt > 1 ? sqrt(fmax((c - a * mean) / (cnt - 1), 0.0)) : NAN;
        o[2] = mx; o[3] = -nmn;
      }
    }
  }
}
extern "C" int trl_sac_losses_f32(const float* q1, const float* q2, const float* tq1, const float* tq2,
                                       const float* logp_next, const float* rew, const float* term, const float* q1n,
                                       const float* q2n, const float* logp, const float* alpha, float gamma, int B,
                                       float* dq1, float* dq2, float* dq1n, float* dq2n, double* sums,
                                       float* alpha_state, float* alpha_out, float target_entropy, float lr, float beta1,
                                       float beta2, float eps, const double* mom_part, int A, double* mom_out,
                                       void* stream) {
  TRL_REQUIRE(B > 0, "empty batch");
  TRL_REQUIRE(q1 && q2 && tq1 && tq2 && logp_next && rew && term && q1n && q2n && logp, "null input");
  TRL_REQUIRE(dq1 && dq2 && dq1n && dq2n && sums, "null output");
  TRL_REQUIRE((alpha_state && alpha_out) || (!alpha_state && alpha), "alpha: a step state + output, or a value");
  TRL_REQUIRE(!mom_part || (mom_out && A > 0), "moments fold: null output / bad A");
  LossExtras x = {alpha_state, alpha_out, target_entropy, lr, beta1, beta2, eps, mom_part, trl_ceil_div(B, 64), A, mom_out};
  hipLaunchKernelGGL(sac_losses_kernel, dim3(1), dim3(SAC_WIDE), 0, (hipStream_t)stream, q1, q2, tq1, tq2, logp_next, rew,
                     term, q1n, q2n, logp, alpha, gamma, B, dq1, dq2, dq1n, dq2n, sums, x);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- DDPG / TD3 losses (deterministic actor-critic)
// TD target r + (1 - d) gamma Q'(s', a') with Q' = tq1 or min(tq1, tq2), MSE of one or two critics and their
// output gradients (ddpg.py:68-73, td3.py:85-97); with qn given also the policy loss -mean(Q(s, pi(s)))
// and its gradient -1/B (ddpg.py:59-62, td3.py:128-130).  sums (double[4]): q1 loss sum, q2 loss sum,
// sum(-qn), sum rewards.
__global__ __launch_bounds__(SAC_THREADS) void detac_losses_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                                   const float* __restrict__ tq1, const float* __restrict__ tq2,
                                                                   const float* __restrict__ rew, const float* __restrict__ term,
                                                                   const float* __restrict__ qn, float gamma, int B,
                                                                   float* __restrict__ dq1, float* __restrict__ dq2,
                                                                   float* __restrict__ dqn, double* __restrict__ sums) {
  __shared__ double smem[SAC_THREADS / 64];
  const float inv_b = 1.0f / (float)B;
  double s1 = 0, s2 = 0, sp = 0, sr = 0;
  for (int b = threadIdx.x; b < B; b += SAC_THREADS) {
    const float tv = tq2 ? fminf(tq1[b], tq2[b]) : tq1[b];
    const float qt = rew[b] + (1.0f - term[b]) * gamma * tv;
    const float e1 = q1[b] - qt;
    dq1[b] = 2.0f * e1 * inv_b;
    s1 += (double)e1 * e1;
    if (q2) { const float e2 = q2[b] - qt; dq2[b] = 2.0f * e2 * inv_b; s2 += (double)e2 * e2; }
    if (qn) { dqn[b] = -inv_b; sp -= (double)qn[b]; }
    sr += (double)rew[b];
  }
  s1 = block_sum(s1, smem); s2 = block_sum(s2, smem); sp = block_sum(sp, smem); sr = block_sum(sr, smem);
  if (threadIdx.x == 0) { sums[0] = s1; sums[1] = s2; sums[2] = sp; sums[3] = sr; }
}
extern "C" int trl_detac_losses_f32(const float* q1, const float* q2, const float* tq1, const float* tq2,
                                    const float* rew, const float* term, const float* qn, float gamma, int B,
                                    float* dq1, float* dq2, float* dqn, double* sums, void* stream) {
  TRL_REQUIRE(B > 0, "empty batch");
  TRL_REQUIRE(q1 && tq1 && rew && term && dq1 && sums, "null pointer");
  TRL_REQUIRE((q2 == nullptr) == (dq2 == nullptr) && (qn == nullptr) == (dqn == nullptr), "q2/dq2 and qn/dqn come in pairs");
  // one workgroup: B is a few thousand and the sums must be order-deterministic
  hipLaunchKernelGGL(detac_losses_kernel, dim3(1), dim3(SAC_THREADS), 0, (hipStream_t)stream, q1, q2, tq1, tq2, rew, term,
                     qn, gamma, B, dq1, dq2, dqn, sums);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// out = clamp(a + clamp(sigma * eps, -noise_clip, noise_clip), lo, hi): exploration noise of
// FixGuassianContPolicy.explore (continuous_policy.py:67-74; noise_clip = lo = -inf.. pass +-inf) and the
// target-policy smoothing of TD3 (td3.py:75-82)
__global__ __launch_bounds__(SAC_THREADS) void noisy_action_kernel(const float* __restrict__ a, const float* __restrict__ eps,
                                                                   float sigma, float noise_clip, float lo, float hi,
                                                                   float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * SAC_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * SAC_THREADS) {
    const float nz = fminf(fmaxf(sigma * eps[e], -noise_clip), noise_clip);
    out[e] = fminf(fmaxf(a[e] + nz, lo), hi);
  }
}
extern "C" int trl_noisy_action_f32(const float* act, const float* eps, float sigma, float noise_clip, float lo, float hi,
                                    float* out, int64_t n, void* stream) {
  TRL_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(act && eps && out, "null pointer");
  int grid = trl_ceil_div(n, SAC_THREADS);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(noisy_action_kernel, dim3(grid), dim3(SAC_THREADS), 0, (hipStream_t)stream, act, eps, sigma,
                     noise_clip, lo, hi, out, n);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- d(policy loss)/d(action): columns [off, off+A) of dx1 + dx2
__global__ __launch_bounds__(SAC_THREADS) void slice_add_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                                float* __restrict__ out, int rows, int ld, int off, int A) {
  const int e = blockIdx.x * SAC_THREADS + threadIdx.x;
  if (e >= rows * A) return;
  const int r = e / A, c = e - r * A;
  out[e] = x1[(size_t)r * ld + off + c] + (x2 ? x2[(size_t)r * ld + off + c] : 0.0f);
}
extern "C" int trl_slice_add_f32(const float* x1, const float* x2, float* out, int rows, int ld, int off, int A,
                                 void* stream) {
  TRL_REQUIRE(rows >= 0 && A > 0 && off >= 0 && off + A <= ld, "bad sizes");
  if (rows == 0) return TRL_OK;
  TRL_REQUIRE(x1 && out, "null pointer");
  hipLaunchKernelGGL(slice_add_kernel, dim3(trl_ceil_div((int64_t)rows * A, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, x1, x2, out, rows, ld, off, A);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- input gradient of a 1-wide head, gated
// out[m][f] = dq[m] * w[f] * act'(h[m][f]): dH = dq w^T of a layer with ONE output is a rank-1 product, and the layer
// below wants it gated by its own activation -- one streaming pass (h read, out written) instead of a K = 1 GEMM plus a
// gate operand in the two GEMMs that consume it.  Up to 12 problems per launch (blockIdx.y).
#define OG_MAX 12
struct OuterGate { const float* dq[OG_MAX]; const float* w[OG_MAX]; const float* h[OG_MAX]; float* out[OG_MAX]; };
__global__ __launch_bounds__(SAC_THREADS) void outer_gate_kernel(OuterGate g, int M, int N4, int act) {
  const int p = blockIdx.y;
  const f32x4* h4 = reinterpret_cast<const f32x4*>(g.h[p]);
  const float* wp = g.w[p];                          // (a parameter view: no alignment promise, and only N floats)
  f32x4* o4 = reinterpret_cast<f32x4*>(g.out[p]);
  const int64_t total = (int64_t)M * N4;
  for (int64_t e = (int64_t)blockIdx.x * SAC_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * SAC_THREADS) {
    const int m = (int)(e / N4), c = (int)(e - (int64_t)m * N4);
    const float d = g.dq[p][m];
    const f32x4 h = h4[e];
    const f32x4 w = {wp[4 * c], wp[4 * c + 1], wp[4 * c + 2], wp[4 * c + 3]};
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ga = act == TRL_ACT_TANH ? 1.0f - h[r] * h[r] : (act == TRL_ACT_RELU ? (h[r] > 0.0f ? 1.0f : 0.0f) : 1.0f);
      o[r] = d * w[r] * ga;
    }
    o4[e] = o;
  }
}
extern "C" int trl_outer_gate_group_f32(int G, const float* const* dq, const float* const* w, const float* const* h,
                                        float* const* out, int M, int N, int act, void* stream) {
  TRL_REQUIRE(G >= 1 && G <= OG_MAX && M >= 0 && N > 0 && (N & 3) == 0, "1..12 problems, N a multiple of 4");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  if (M == 0) return TRL_OK;
  TRL_REQUIRE(dq && w && h && out, "null pointer array");
  OuterGate g;
  for (int k = 0; k < G; ++k) {
    TRL_REQUIRE(dq[k] && w[k] && h[k] && out[k], "null pointer");
    TRL_REQUIRE(((reinterpret_cast<uintptr_t>(h[k]) | reinterpret_cast<uintptr_t>(out[k])) & 15) == 0,
                "h / out must be 16-byte aligned");
    g.dq[k] = dq[k]; g.w[k] = w[k]; g.h[k] = h[k]; g.out[k] = out[k];
  }
  int grid = trl_ceil_div((int64_t)M * (N / 4), SAC_THREADS);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(outer_gate_kernel, dim3(grid, G), dim3(SAC_THREADS), 0, (hipStream_t)stream, g, M, N / 4, act);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- K13 Polyak target update
// target <- (1 - tau) target + tau source   (atu.soft_update_from_to, torchrl/algo/utils.py:16-20)
__global__ __launch_bounds__(SAC_THREADS) void polyak_kernel(float* __restrict__ tgt, const float* __restrict__ src,
                                                             int64_t n, float tau) {
  for (int64_t e = (int64_t)blockIdx.x * SAC_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * SAC_THREADS)
    tgt[e] = tgt[e] * (1.0f - tau) + src[e] * tau;
}
extern "C" int trl_polyak_f32(float* target, const float* source, int64_t n, float tau, void* stream) {
  TRL_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(target && source, "null pointer");
  int grid = trl_ceil_div(n, SAC_THREADS);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(polyak_kernel, dim3(grid), dim3(SAC_THREADS), 0, (hipStream_t)stream, target, source, n, tau);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- mean / unbiased std / max / min of a tensor (logging)
#define MOM_THREADS 1024
__device__ __forceinline__ void moments_block(const float* __restrict__ x, int64_t n, int ld, int off,
                                              int width, float lo, float hi_, double* __restrict__ out,
                                              double* __restrict__ out2 = nullptr) {
  // x viewed as rows of `ld` floats; statistics over columns [off, off+width) of every row.  One workgroup of 16
  // waves (fixed summation order); each thread walks (row, column) incrementally -- no division in the loop.
  __shared__ double smem[4][MOM_THREADS / 64];
  double s = 0, sq = 0, mx = -INFINITY, nmn = -INFINITY;
  const int64_t rows = n / ld;
  // four independent element streams per thread (elements tid + j * 1024, then strides of 4096): their loads are in flight
  // together -- one stream per thread was a chain of dependent round trips, 24 of them at B = 4096 x 6 columns
  int64_t r[4];
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = threadIdx.x + j * MOM_THREADS;
    r[j] = e / width;
    c[j] = e - (int)r[j] * width;
  }
  const int dr = 4 * MOM_THREADS / width, dc = 4 * MOM_THREADS - dr * width;
  while (r[0] < rows) {                                                // (stream 0 is the one that runs out last)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = r[j] < rows ? x[r[j] * ld + off + c[j]] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r[j] < rows) {
        const double d = (double)fminf(fmaxf(v[j], lo), hi_);
        s += d; sq += d * d; mx = fmax(mx, d); nmn = fmax(nmn, -d);
      }
      r[j] += dr; c[j] += dc;
      if (c[j] >= width) { c[j] -= width; ++r[j]; }
    }
  }
  s = wave_sum(s); sq = wave_sum(sq); mx = wave_max(mx); nmn = wave_max(nmn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { smem[0][wave] = s; smem[1][wave] = sq; smem[2][wave] = mx; smem[3][wave] = nmn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = 0; sq = 0; mx = -INFINITY; nmn = -INFINITY;
    for (int w = 0; w < MOM_THREADS / 64; ++w) {
      s += smem[0][w]; sq += smem[1][w]; mx = fmax(mx, smem[2][w]); nmn = fmax(nmn, smem[3][w]);
    }
    const double cnt = (double)(rows * width), mean = s / cnt;
    const double sd = cnt > 1 ? sqrt(fmax((sq - s * mean) / (cnt - 1), 0.0)) : NAN;
    out[0] = mean; out[1] = sd; out[2] = mx; out[3] = -nmn;
    if (out2) { out2[0] = mean; out2[1] = sd; out2[2] = mx; out2[3] = -nmn; }      // (the ring slot's copy)
  }
}
// up to 4 of the above in one launch (blockIdx.x = statistic): the three logged tensors of a SAC update
#define MOM_MAX 4
struct MomSet { const float* x[MOM_MAX]; int64_t n[MOM_MAX]; int ld[MOM_MAX], off[MOM_MAX], width[MOM_MAX];
                float lo[MOM_MAX], hi[MOM_MAX]; double* out[MOM_MAX]; };
// `ring` (optional): the statistics block `raw` (which holds every out[k]) is also filed into slot (update count - 1) mod
// slots of a device ring -- the last launch of an update archives its logged numbers, no copy command per update.
struct MomRing { const uint8_t* raw; uint8_t* ring; const double* step; int raw_bytes, slots; };
__global__ __launch_bounds__(MOM_THREADS) void moments_multi_kernel(MomSet m, MomRing r, int count) {
  const int k = blockIdx.x;
  if (!r.ring) {
    moments_block(m.x[k], m.n[k], m.ld[k], m.off[k], m.width[k], m.lo[k], m.hi[k], m.out[k]);
    return;
  }
  const int64_t done = (int64_t)r.step[0] - 1;                         // (the step state was advanced earlier in the update)
  uint8_t* slot = r.ring + (int64_t)(((done % r.slots) + r.slots) % r.slots) * r.raw_bytes;
  if (k == 0)                                                          // everything earlier launches left in `raw`
    for (int w = threadIdx.x; w < r.raw_bytes / 4; w += MOM_THREADS) {
      const uint8_t* at = r.raw + 4 * w;
      bool mine = true;
      for (int j = 0; j < count; ++j) {
        const uint8_t* o = reinterpret_cast<const uint8_t*>(m.out[j]);
        if (at >= o && at < o + 32) mine = false;
      }
      if (mine) reinterpret_cast<uint32_t*>(slot)[w] = reinterpret_cast<const uint32_t*>(r.raw)[w];
    }
  moments_block(m.x[k], m.n[k], m.ld[k], m.off[k], m.width[k], m.lo[k], m.hi[k], m.out[k],   // this block's own four numbers
                reinterpret_cast<double*>(slot + (reinterpret_cast<const uint8_t*>(m.out[k]) - r.raw)));
}
static int moments_multi(int count, const float* const* x, const int64_t* n, const int* ld, const int* off,
                         const int* width, const float* clamp_lo, const float* clamp_hi, double* const* out4,
                         MomRing ring, void* stream) {
  TRL_REQUIRE(count >= 1 && count <= MOM_MAX, "moments_multi: 1..4 statistics");
  TRL_REQUIRE(x && n && ld && off && width && clamp_lo && clamp_hi && out4, "null pointer");
  MomSet m;
  for (int k = 0; k < count; ++k) {
    TRL_REQUIRE(n[k] > 0 && ld[k] > 0 && off[k] >= 0 && width[k] > 0 && off[k] + width[k] <= ld[k] && n[k] % ld[k] == 0, "bad sizes");
    TRL_REQUIRE(x[k] && out4[k], "null pointer");
    m.x[k] = x[k]; m.n[k] = n[k]; m.ld[k] = ld[k]; m.off[k] = off[k]; m.width[k] = width[k];
    m.lo[k] = clamp_lo[k]; m.hi[k] = clamp_hi[k]; m.out[k] = out4[k];
  }
  if (ring.ring)
    for (int k = 0; k < count; ++k) {
      const uint8_t* o = reinterpret_cast<const uint8_t*>(out4[k]);
      TRL_REQUIRE(o >= ring.raw && o + 32 <= ring.raw + ring.raw_bytes, "moments_multi_ring: out4 outside the statistics block");
    }
  hipLaunchKernelGGL(moments_multi_kernel, dim3(count), dim3(MOM_THREADS), 0, (hipStream_t)stream, m, ring, count);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
// `count` (1..4) statistics in one launch; with `ring` (nullable) the launch also files the update's statistics block
// `raw` into slot ((int64)update_count[0] - 1) mod slots
extern "C" int trl_moments_multi_f64(int count, const float* const* x, const int64_t* n, const int* ld, const int* off,
                                     const int* width, const float* clamp_lo, const float* clamp_hi, double* const* out4,
                                     const void* raw, int raw_bytes, void* ring, int slots, const double* update_count,
                                     void* stream) {
  if (!ring)
    return moments_multi(count, x, n, ld, off, width, clamp_lo, clamp_hi, out4, MomRing{nullptr, nullptr, nullptr, 0, 0}, stream);
  TRL_REQUIRE(raw && update_count && raw_bytes > 0 && raw_bytes % 8 == 0 && slots > 0, "moments_multi: bad ring");
  return moments_multi(count, x, n, ld, off, width, clamp_lo, clamp_hi, out4,
                       MomRing{(const uint8_t*)raw, (uint8_t*)ring, update_count, raw_bytes, slots}, stream);
}

// ---------------------------------------------------------------- N(0,1) fill from the Philox stream
// element e uses counter (c0 = lo32(ctr), c1 = hi32(ctr), block = e / 4, tag NOISE) under `seed`
__global__ __launch_bounds__(SAC_THREADS) void philox_normal_kernel(float* __restrict__ out, int64_t n, int64_t seed,
                                                                    int64_t ctr) {
  const int64_t blk = (int64_t)blockIdx.x * SAC_THREADS + threadIdx.x;
  if (blk * 4 >= n) return;
  float z[4];
  philox_normals4((uint32_t)(ctr & 0xFFFFFFFFll), (uint32_t)((ctr >> 32) & 0xFFFFFFFFll), (uint32_t)blk, TRL_TAG_NOISE,
                  seed, z);
  for (int c = 0; c < 4; ++c) if (blk * 4 + c < n) out[blk * 4 + c] = z[c];
}
extern "C" int trl_philox_normal_f32(float* out, int64_t n, int64_t seed, int64_t counter, void* stream) {
  TRL_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(out, "null pointer");
  hipLaunchKernelGGL(philox_normal_kernel, dim3(trl_ceil_div((n + 3) / 4, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, out, n, seed, counter);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- K1 stand-alone: one vector step of the synthetic env
// VecEnv.step (torchrl/env/vecenv.py:53-61) for SynthVecEnv: obs' = tanh(obs A + act B),
// reward = scale (obs'[0] - 0.1 |act|^2), done = time_limit = (t_env + 1 >= horizon).
// Writes next_obs into cur_obs (in place) AND into next_out; rewards / dones as (N) floats.
__global__ __launch_bounds__(SAC_THREADS) void synth_step_kernel(float* __restrict__ cur_obs, const float* __restrict__ act,
                                                                 const float* __restrict__ envA, const float* __restrict__ envB,
                                                                 int32_t* __restrict__ t_env, float reward_scale, int horizon,
                                                                 float* __restrict__ next_out, float* __restrict__ rew,
                                                                 float* __restrict__ done, int N, int D, int A) {
  extern __shared__ float sm[];                    // envA (D*D) | envB (A*D)
  for (int e = threadIdx.x; e < D * D; e += SAC_THREADS) sm[e] = envA[e];
  for (int e = threadIdx.x; e < A * D; e += SAC_THREADS) sm[D * D + e] = envB[e];
  __syncthreads();
  const int n = blockIdx.x * SAC_THREADS + threadIdx.x;
  if (n >= N) return;
  float o[32], a[8];
  for (int k = 0; k < D; ++k) o[k] = cur_obs[(size_t)n * D + k];
  float asq = 0.0f;
  for (int k = 0; k < A; ++k) { a[k] = act[(size_t)n * A + k]; asq = fmaf(a[k], a[k], asq); }
  float first = 0.0f;
  for (int f = 0; f < D; ++f) {
    float p = 0.0f;
    for (int k = 0; k < D; ++k) p = fmaf(o[k], sm[k * D + f], p);
    for (int k = 0; k < A; ++k) p = fmaf(a[k], sm[D * D + k * D + f], p);
    const float v = trl_tanh(p);
    if (f == 0) first = v;
    cur_obs[(size_t)n * D + f] = v;
    next_out[(size_t)n * D + f] = v;
  }
  const int t = t_env[n] + 1;
  t_env[n] = t;
  rew[n] = reward_scale * (first - 0.1f * asq);
  done[n] = t >= horizon ? 1.0f : 0.0f;
}
extern "C" int trl_synth_env_step_f32(float* cur_obs, const float* act, const float* env_A, const float* env_B,
                                      int32_t* t_env, float reward_scale, int horizon, float* next_obs, float* rewards,
                                      float* dones, int N, int D, int A, void* stream) {
  TRL_REQUIRE(N >= 0 && D > 0 && D <= 32 && A > 0 && A <= 8, "bad sizes (D <= 32, A <= 8)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(cur_obs && act && env_A && env_B && t_env && next_obs && rewards && dones, "null pointer");
  hipLaunchKernelGGL(synth_step_kernel, dim3(trl_ceil_div(N, SAC_THREADS)), dim3(SAC_THREADS),
                     (D * D + A * D) * sizeof(float), (hipStream_t)stream, cur_obs, act, env_A, env_B, t_env,
                     reward_scale, horizon, next_obs, rewards, dones, N, D, A);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- one off-policy vector step on the synthetic env, ONE launch
// VecCollector.take_actions (torchrl/collector/base.py:184-230) for a GuassianContPolicy on the on-GPU vector env, after
// the policy MLP: sample the action from the head (rsample_fwd_kernel), store obs / acts, env.step (synth_step_kernel),
// store next_obs / rewards / terminals / time_limits, the collector's bookkeeping (collector_bookkeep_kernel) and the
// partial reset of finished envs (synth_reset_kernel + synth_bump_episode_kernel) -- eight launches of a few microseconds
// of work each, host-launch-bound at ~10 us apiece.  Same arithmetic, same Philox blocks.
struct CollectStep {
  float* cur_obs; const float* head; const float* eps; const float* envA; const float* envB;
  int64_t noise_seed, noise_ctr; int noise_row0;   // eps == NULL: the exploration noise of trl_philox_normal_f32(seed, ctr),
                                                   // rows [noise_row0, noise_row0 + N) of the (all envs, A) draw, made in place
  int32_t* t_env; int32_t* cur_step; int32_t* episode_idx; float* ep_return;
  float reward_scale; int horizon, max_frames; int64_t seed_base;
  float* obs_row; float* acts_row; float* next_row; float* rew_row; float* done_row; float* tl_row;   // obs / acts / tl nullable
  uint8_t* mask; double* epoch_reward; int32_t* ep_count; float* ep_log; int ep_cap, step;
  int N, D, A, tanh_action;
  // graph-replayable form (dyn != NULL): the step counter, the ring row and the epoch's first step live on the device --
  // dyn = {global step, ring row, first step of the epoch}; the row pointers above are then the BASES of the ring
  // tensors (n_rows rows); the block that retires last advances dyn[0] and dyn[1] (done: its counter, zero between launches)
  int64_t* dyn; int n_rows; unsigned* done;
};
// Workgroup = 8 envs x 32 feature slots (D <= 32, A <= 8): the observation and the action of an env sit in LDS, lane f of
// the env's 32 computes feature f of the next observation (the same dot products in the same order as one thread per env
// made them -- bit-identical -- but 17 of them side by side and the rows written as contiguous runs), lane 0 does the
// env's bookkeeping.  One thread per env was 4 workgroups at N = 1024 walking 391 dependent FMAs and ~100 stores 68
// bytes apart each: 28 us per vector step.
#define CS_EPB (SAC_THREADS / 32)
__global__ __launch_bounds__(SAC_THREADS) void synth_collect_step_kernel(CollectStep c) {
  extern __shared__ float sm[];                    // envA (D*D) | envB (A*D)
  __shared__ double red[SAC_THREADS / 64];
  __shared__ float so[CS_EPB][32], sa[CS_EPB][8];
  __shared__ int sflag[CS_EPB], sep[CS_EPB];
  const int D = c.D, A = c.A;
  // (locals, not writes into `c`: a modified kernel-argument struct is demoted to scratch memory)
  int64_t dyn_gs = 0, dyn_row = 0, noise_ctr = c.noise_ctr;
  int step = c.step;
  float* obs_row = c.obs_row; float* acts_row = c.acts_row; float* next_row = c.next_row;
  float* rew_row = c.rew_row; float* done_row = c.done_row; float* tl_row = c.tl_row;
  if (c.dyn) {                                     // uniform: every block reads the state before any block can retire
    dyn_gs = c.dyn[0]; dyn_row = c.dyn[1];
    noise_ctr = dyn_gs; step = (int)(dyn_gs - c.dyn[2]);
    const size_t r = (size_t)dyn_row * c.N;
    if (obs_row) obs_row += r * D;
    if (acts_row) acts_row += r * A;
    next_row += r * D; rew_row += r; done_row += r;
    if (tl_row) tl_row += r;
  }
  for (int e = threadIdx.x; e < D * D; e += SAC_THREADS) sm[e] = c.envA[e];
  for (int e = threadIdx.x; e < A * D; e += SAC_THREADS) sm[D * D + e] = c.envB[e];
  const int le = threadIdx.x >> 5, f = threadIdx.x & 31;
  const int n = blockIdx.x * CS_EPB + le;
  const bool live = n < c.N;
  if (live && f < D) so[le][f] = c.cur_obs[(size_t)n * D + f];
  if (live && f < A) {                             // action element f of the env: TanhNormal sample from the head
    float ez;
    if (c.eps) {
      ez = c.eps[(size_t)n * A + f];
    } else {                                       // element e of the draw = normal (e & 3) of Philox block e / 4
      const int64_t e = ((int64_t)c.noise_row0 + n) * A + f;
      float z[4];
      philox_normals4((uint32_t)(noise_ctr & 0xFFFFFFFFll), (uint32_t)((noise_ctr >> 32) & 0xFFFFFFFFll),
                      (uint32_t)(e >> 2), TRL_TAG_NOISE, c.noise_seed, z);
      const int q = (int)(e & 3);
      ez = q == 0 ? z[0] : (q == 1 ? z[1] : (q == 2 ? z[2] : z[3]));
    }
    const float* head = c.head + (size_t)n * 2 * A;
    const float mean = head[f];
    const float sd = __expf(fminf(fmaxf(head[A + f], -20.0f), 2.0f));
    const float z = fmaf(sd, ez, mean);
    const float a = c.tanh_action ? trl_tanh(z) : z;
    sa[le][f] = a;
    if (acts_row) acts_row[(size_t)n * A + f] = a;
  }
  __syncthreads();
  float nv = 0.0f;
  if (live && f < D) {                             // obs' = tanh(obs A + act B), feature f
    if (obs_row) obs_row[(size_t)n * D + f] = so[le][f];
    float p = 0.0f;
    for (int k = 0; k < D; ++k) p = fmaf(so[le][k], sm[k * D + f], p);
    for (int k = 0; k < A; ++k) p = fmaf(sa[le][k], sm[D * D + k * D + f], p);
    nv = trl_tanh(p);
    next_row[(size_t)n * D + f] = nv;
  }
  double r = 0.0;
  int t = 0;
  if (live && f == 0) {
    float asq = 0.0f;
    for (int k = 0; k < A; ++k) asq = fmaf(sa[le][k], sa[le][k], asq);
    t = c.t_env[n] + 1;
    const float rew = c.reward_scale * (nv - 0.1f * asq);
    const bool d = t >= c.horizon;
    rew_row[n] = rew; done_row[n] = d ? 1.0f : 0.0f;
    if (tl_row) tl_row[n] = d ? 1.0f : 0.0f;       // synthetic env: time_limit == done
    r = (double)rew;
    const int cs = c.cur_step[n] + 1;              // ---- collector bookkeeping ----
    float er = c.ep_return[n] + rew;
    if (d) {
      const int slot = atomicAdd(c.ep_count, 1);
      if (slot < c.ep_cap) { c.ep_log[slot * 3 + 0] = (float)step; c.ep_log[slot * 3 + 1] = (float)n; c.ep_log[slot * 3 + 2] = er; }
      er = 0.0f;
    }
    const bool flag = d || cs >= c.max_frames;
    c.cur_step[n] = flag ? 0 : cs;
    c.ep_return[n] = er;
    c.mask[n] = flag ? 1 : 0;
    const int ep = c.episode_idx[n] + 1;
    sflag[le] = flag ? 1 : 0; sep[le] = ep;
    c.t_env[n] = flag ? 0 : t;
    if (flag) c.episode_idx[n] = ep;
  }
  __syncthreads();
  if (live && f < D) {
    if (sflag[le]) {                               // ---- partial reset: a fresh Philox observation (episode counter + 1) ----
      float z[4];
      philox_normals4((uint32_t)sep[le], 0u, (uint32_t)(f >> 2), TRL_TAG_RESET, c.seed_base + n, z);
      const int q = f & 3;
      c.cur_obs[(size_t)n * D + f] = q == 0 ? z[0] : (q == 1 ? z[1] : (q == 2 ? z[2] : z[3]));
    } else {
      c.cur_obs[(size_t)n * D + f] = nv;
    }
  }
  r = block_sum(r, red);
  if (threadIdx.x == 0 && c.epoch_reward) atomicAdd(c.epoch_reward, r);
  if (c.dyn && threadIdx.x == 0) {                 // (block_sum's barriers: every thread of this block is done with dyn)
    const unsigned before = __hip_atomic_fetch_add(c.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {
      c.dyn[0] = dyn_gs + 1; c.dyn[1] = dyn_row + 1 == c.n_rows ? 0 : dyn_row + 1;
      __hip_atomic_store(c.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// `state` NULL: one step with host-side bookkeeping -- the six row pointers are THE rows to write (obs / acts / tl may be
// NULL: evaluation stores nothing), eps is read (NULL: drawn in place for (noise_seed, noise_counter)), `step` is the
// logged step.  `state` given: every per-step quantity is on the device, so that the launch (and the policy pass in front
// of it) is captured once and replayed for every vector step: state = {global step, ring row, first step of the epoch}
// (3 int64) followed by a zeroed 32-bit block counter at state + 3; the six pointers are then the whole ring tensors
// (n_rows time rows of N envs), the noise is drawn in place (counter = global step), the launch stores into row state[1]
// and advances state[0] and state[1] itself.
extern "C" int trl_synth_collect_step_f32(float* cur_obs, const float* head, const float* eps, int64_t noise_seed,
                                          int64_t noise_counter, int noise_row0, const float* env_A,
                                          const float* env_B, int32_t* t_env, int32_t* cur_step, int32_t* episode_idx,
                                          float* ep_return, float reward_scale, int horizon, int max_episode_frames,
                                          int64_t env_seed_base, float* obs, float* acts, float* next_obs,
                                          float* rewards, float* terminals, float* time_limits, int n_rows, int64_t* state,
                                          uint8_t* reset_mask, double* epoch_reward, int32_t* ep_count, float* ep_log,
                                          int ep_cap, int step, int N, int D, int A, int tanh_action, void* stream) {
  TRL_REQUIRE(N >= 0 && D > 0 && D <= 32 && A > 0 && A <= 8 && ep_cap >= 0, "bad sizes (D <= 32, A <= 8)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(cur_obs && head && env_A && env_B && t_env && cur_step && episode_idx && ep_return, "null pointer");
  TRL_REQUIRE(next_obs && rewards && terminals && reset_mask && ep_count && ep_log && noise_row0 >= 0, "null pointer");
  TRL_REQUIRE(!state || (obs && acts && n_rows > 0 && !eps), "device-side step state: whole ring tensors, noise drawn in place");
  CollectStep c{cur_obs, head, eps, env_A, env_B, noise_seed, state ? 0 : noise_counter, noise_row0, t_env, cur_step,
                episode_idx, ep_return, reward_scale, horizon, max_episode_frames, env_seed_base, obs, acts, next_obs,
                rewards, terminals, time_limits, reset_mask, epoch_reward, ep_count, ep_log, ep_cap, state ? 0 : step, N, D,
                A, tanh_action, state, state ? n_rows : 0, state ? reinterpret_cast<unsigned*>(state + 3) : nullptr};
  hipLaunchKernelGGL(synth_collect_step_kernel, dim3(trl_ceil_div(N, CS_EPB)), dim3(SAC_THREADS),
                     (D * D + A * D) * sizeof(float), (hipStream_t)stream, c);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- off-policy collector bookkeeping
// VecCollector.take_actions after env.step (torchrl/collector/base.py:205-224): step counters,
// running returns (logged and cleared on done only), reset mask = done | step >= max_episode_frames.
__global__ __launch_bounds__(SAC_THREADS) void collector_bookkeep_kernel(const float* __restrict__ rew,
                                                                         const float* __restrict__ done,
                                                                         int32_t* __restrict__ cur_step,
                                                                         float* __restrict__ ep_return, int max_frames,
                                                                         uint8_t* __restrict__ mask,
                                                                         double* __restrict__ epoch_reward,
                                                                         int32_t* __restrict__ ep_count,
                                                                         float* __restrict__ ep_log, int ep_cap, int step,
                                                                         int N) {
  __shared__ double smem[SAC_THREADS / 64];
  const int n = blockIdx.x * SAC_THREADS + threadIdx.x;
  double r = 0.0;
  if (n < N) {
    r = (double)rew[n];
    const bool d = done[n] != 0.0f;
    const int cs = cur_step[n] + 1;
    float er = ep_return[n] + rew[n];
    if (d) {
      const int slot = atomicAdd(ep_count, 1);
      if (slot < ep_cap) { ep_log[slot * 3 + 0] = (float)step; ep_log[slot * 3 + 1] = (float)n; ep_log[slot * 3 + 2] = er; }
      er = 0.0f;
    }
    const bool flag = d || cs >= max_frames;
    cur_step[n] = flag ? 0 : cs;
    ep_return[n] = er;
    mask[n] = flag ? 1 : 0;
  }
  r = block_sum(r, smem);
  if (threadIdx.x == 0 && epoch_reward) atomicAdd(epoch_reward, r);
}
extern "C" int trl_collector_bookkeep_f32(const float* rewards, const float* dones, int32_t* cur_step, float* ep_return,
                                          int max_episode_frames, uint8_t* reset_mask, double* epoch_reward,
                                          int32_t* ep_count, float* ep_log, int ep_cap, int step, int N, void* stream) {
  TRL_REQUIRE(N >= 0 && ep_cap >= 0, "bad sizes");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(rewards && dones && cur_step && ep_return && reset_mask && ep_count && ep_log, "null pointer");
  hipLaunchKernelGGL(collector_bookkeep_kernel, dim3(trl_ceil_div(N, SAC_THREADS)), dim3(SAC_THREADS), 0,
                     (hipStream_t)stream, rewards, dones, cur_step, ep_return, max_episode_frames, reset_mask,
                     epoch_reward, ep_count, ep_log, ep_cap, step, N);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
