// K7 / K8 for ARBITRARY network shapes -- the loss half of PPO.update / A2C.update as stand-alone kernels
// (reference: torchrl/algo/on_policy/ppo.py:41-152, a2c.py:29-106).
//
// The fused gradient kernel (k_ppo.hip) is instantiated for the benchmark network (17-64-64-{6,1}); any other
// MLP -- other observation / action sizes, other widths or depths -- runs its layers on the generic GEMM family
// (k_gemm.hip) and needs only what sits between the two networks' forward and backward passes:
//   inputs   mean (B, A) and v (B) of this minibatch, the stored acts / advs / returns / old values / log pi_old,
//            the state-independent logstd (A), the minibatch's advantage statistics (trl_adv_stats_f64)
//   outputs  d(loss)/d(mean) (B, A), d(value loss)/d(v) (B), d(loss)/d(logstd) (A), and the 24 statistics of
//            trl_ppo_reduce_f32's info layout
// with exactly the per-sample arithmetic of the fused kernel (same helpers, same clip / tie conventions).
// Two passes: every block writes partial sums, one block folds them in fixed order (deterministic).
#include "trl_common.h"
#include "trl_mlp.h"

#define PG_THREADS 256
#define PG_MAX_A 64
#define PG_SCAL 12          // lp sum, lp^2, max lp, -min lp, max ratio, -min ratio, surrogate sum | vloss, v sum, v^2, max v, -min v

struct PpoGenDev {
  const float* mean; const float* logstd; const float* acts; const float* advs; const float* old_logp;
  const float* v; const float* rets; const float* v_old;
  const double* adv_raw;
  float* d_mean; float* d_v; double* partial;       // partial: [blocks][A + PG_SCAL]
  int B, A;
  float clip_para, entropy_coeff;
  int clipped_value_loss, tanh_action, loss_mode;
  double n_global;
};

__device__ __forceinline__ double pg_block_reduce(double v, bool is_max, double* smem) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = is_max ? -INFINITY : 0.0;
  for (int w = 0; w < PG_THREADS / 64; ++w) r = is_max ? fmax(r, smem[w]) : r + smem[w];
  return r;
}

__global__ __launch_bounds__(PG_THREADS) void ppo_generic_losses_kernel(PpoGenDev a) {
  __shared__ double smem[PG_THREADS / 64];
  __shared__ float s_dls[PG_THREADS / 64][PG_MAX_A];
  const int b = blockIdx.x * PG_THREADS + threadIdx.x;
  const bool valid = b < a.B;
  const int A = a.A;
  // advantage normalisation constants (ppo.py:141-147): mean, unbiased std
  const double ng = a.n_global;
  const double adv_mean = a.adv_raw[0] / ng;
  const double adv_var = (a.adv_raw[1] - a.adv_raw[0] * a.adv_raw[0] / ng) / (ng - 1.0);
  const float adv_mu = (float)adv_mean;
  const float adv_rstd = 1.0f / ((float)sqrt(fmax(adv_var, 0.0)) + 1e-5f);
  const float inv_b = (float)(1.0 / ng);

  // ---- policy: log pi, surrogate, d/d(mean), d/d(logstd) ----
  float lp = 0.0f;
  if (valid)
    for (int o = 0; o < A; ++o) {
      const float ls = fminf(fmaxf(a.logstd[o], -20.0f), 2.0f);
      float zc;
      lp += gauss_logp_term(a.acts[(size_t)b * A + o], a.mean[(size_t)b * A + o], __expf(-2.0f * ls), ls, a.tanh_action, zc);
    }
  const float advn = valid ? (a.advs[b] - adv_mu) * adv_rstd : 0.0f;
  float ratio, s1, s2, g_lp;
  if (a.loss_mode == TRL_LOSS_A2C) {                              // L = -mean(log pi * adv) (a2c.py:69-70)
    ratio = 1.0f;
    s1 = s2 = lp * advn;
    g_lp = valid ? -advn * inv_b : 0.0f;
  } else {                                                       // clipped surrogate (ppo.py:58-66)
    ratio = valid ? __expf(lp - a.old_logp[b]) : 1.0f;
    s1 = ratio * advn;
    s2 = fminf(fmaxf(ratio, 1.0f - a.clip_para), 1.0f + a.clip_para) * advn;
    g_lp = (valid && s1 <= s2) ? -advn * ratio * inv_b : 0.0f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 0; o < A; ++o) {
    float dls = 0.0f;
    if (valid) {
      const float raw = a.logstd[o];
      const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
      const float pass = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
      const float ivv = __expf(-2.0f * ls);
      float zc;
      gauss_logp_term(a.acts[(size_t)b * A + o], a.mean[(size_t)b * A + o], ivv, ls, a.tanh_action, zc);
      a.d_mean[(size_t)b * A + o] = g_lp * zc * ivv;
      dls = pass * (g_lp * (zc * zc * ivv - 1.0f) - a.entropy_coeff * inv_b);
    }
    dls = wave_sum(dls);
    if (lane == 0) s_dls[wave][o] = dls;
  }
  // ---- value: loss and d/d(v) ----
  float vv = 0.0f, l = 0.0f;
  if (valid) {
    vv = a.v[b];
    const float R = a.rets[b];
    float dv;
    if (a.clipped_value_loss) {                                  // ppo.py:104-111
      const float vo = a.v_old[b];
      const float dc = vv - vo;
      const float vc = vo + fminf(fmaxf(dc, -a.clip_para), a.clip_para);
      const float l1 = (vv - R) * (vv - R), l2 = (vc - R) * (vc - R);
      const float wa = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), wb = 1.0f - wa;
      const float pass = (dc >= -a.clip_para && dc <= a.clip_para) ? 1.0f : 0.0f;
      l = 0.5f * fmaxf(l1, l2);
      dv = inv_b * (wa * (vv - R) + wb * pass * (vc - R));
    } else {                                                     // nn.MSELoss, a2c.py:43
      l = (vv - R) * (vv - R);
      dv = 2.0f * (vv - R) * inv_b;
    }
    a.d_v[b] = dv;
  }
  // ---- block partials ----
  double* out = a.partial + (size_t)blockIdx.x * (A + PG_SCAL);
  __syncthreads();
  if (threadIdx.x < A) {
    float s = 0.0f;
    for (int w = 0; w < PG_THREADS / 64; ++w) s += s_dls[w][threadIdx.x];
    out[threadIdx.x] = (double)s;
  }
  const double ninf = -INFINITY;
  const double vals[PG_SCAL] = {valid ? (double)lp : 0.0, valid ? (double)lp * lp : 0.0, valid ? (double)lp : ninf,
                                valid ? -(double)lp : ninf, valid ? (double)ratio : ninf, valid ? -(double)ratio : ninf,
                                valid ? -(double)fminf(s1, s2) : 0.0, (double)l, (double)vv, (double)vv * vv,
                                valid ? (double)vv : ninf, valid ? -(double)vv : ninf};
  const bool is_max[PG_SCAL] = {false, false, true, true, true, true, false, false, false, false, true, true};
#pragma unroll
  for (int k = 0; k < PG_SCAL; ++k) {
    const double r = pg_block_reduce(vals[k], is_max[k], smem);
    if (threadIdx.x == 0) out[A + k] = r;
  }
}

// one block: fold the block partials in order; d_logstd (A) and the info row (trl_ppo_reduce_f32's layout)
__global__ __launch_bounds__(PG_THREADS) void ppo_generic_fold_kernel(const double* __restrict__ partial, int blocks, int A,
                                                                    const float* __restrict__ logstd,
                                                                    float* __restrict__ d_logstd, double* __restrict__ info) {
  __shared__ double s_out[PG_MAX_A + PG_SCAL];
  const int stride = A + PG_SCAL;
  for (int e = threadIdx.x; e < stride; e += PG_THREADS) {
    const int k = e - A;
    const bool is_max = k >= 0 && (k == 2 || k == 3 || k == 4 || k == 5 || k == 10 || k == 11);
    double r = is_max ? -INFINITY : 0.0;
    for (int w = 0; w < blocks; ++w) {
      const double o = partial[(size_t)w * stride + e];
      r = is_max ? fmax(r, o) : r + o;
    }
    s_out[e] = r;
    if (e < A) d_logstd[e] = (float)r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double* s = s_out + A;
    info[0] = s[6]; info[1] = s[0]; info[2] = s[1]; info[3] = s[2]; info[4] = s[3]; info[5] = s[4]; info[6] = s[5];
    info[7] = s[7]; info[12] = s[8]; info[13] = s[9]; info[14] = s[10]; info[15] = s[11];
    double sm = 0, sq = 0, mx = -INFINITY, mn = INFINITY, es = 0, eq = 0, emx = -INFINITY, emn = INFINITY;
    for (int o = 0; o < A; ++o) {                                // ppo.py:82-85 log_std/*, a2c.py:95-100 std/*
      const double x = fmin(fmax((double)logstd[o], -20.0), 2.0), ex = exp(x);
      sm += x; sq += x * x; mx = fmax(mx, x); mn = fmin(mn, x);
      es += ex; eq += ex * ex; emx = fmax(emx, ex); emn = fmin(emn, ex);
    }
    const double mean = sm / A, em = es / A;
    info[8] = mean; info[9] = A > 1 ? sqrt(fmax((sq - sm * mean) / (A - 1), 0.0)) : NAN; info[10] = mx; info[11] = mn;
    info[16] = em; info[17] = A > 1 ? sqrt(fmax((eq - es * em) / (A - 1), 0.0)) : NAN; info[18] = emx; info[19] = emn;
  }
}

extern "C" int trl_ppo_generic_losses_workspace(int B, int A) {
  if (B <= 0 || A <= 0 || A > PG_MAX_A) return TRL_EINVAL;
  return trl_ceil_div(B, PG_THREADS) * (A + PG_SCAL);            // doubles
}

extern "C" int trl_ppo_generic_losses_f32(const float* mean, const float* logstd, const float* acts, const float* advs,
                                          const float* old_logp, const float* v, const float* rets, const float* v_old,
                                          const double* adv_raw, double n_global, int B, int A, float clip_para,
                                          float entropy_coeff, int clipped_value_loss, int tanh_action, int loss_mode,
                                          float* d_mean, float* d_v, float* d_logstd, double* info, double* workspace,
                                          void* stream) {
  TRL_REQUIRE(B > 0 && A > 0 && A <= PG_MAX_A, "bad sizes (1 <= A <= 64)");
  TRL_REQUIRE(mean && logstd && acts && advs && v && rets && adv_raw && d_mean && d_v && d_logstd && info && workspace,
              "null pointer");
  TRL_REQUIRE(loss_mode == TRL_LOSS_A2C || old_logp, "the clipped surrogate needs old_logp");
  TRL_REQUIRE(!clipped_value_loss || v_old, "the clipped value loss needs the old values");
  TRL_REQUIRE(n_global >= 2.0, "need at least two samples for the advantage statistics");
  PpoGenDev a{};
  a.mean = mean; a.logstd = logstd; a.acts = acts; a.advs = advs; a.old_logp = old_logp; a.v = v; a.rets = rets;
  a.v_old = v_old; a.adv_raw = adv_raw; a.d_mean = d_mean; a.d_v = d_v; a.partial = workspace; a.B = B; a.A = A;
  a.clip_para = clip_para; a.entropy_coeff = entropy_coeff; a.clipped_value_loss = clipped_value_loss;
  a.tanh_action = tanh_action; a.loss_mode = loss_mode; a.n_global = n_global;
  const int blocks = trl_ceil_div(B, PG_THREADS);
  hipLaunchKernelGGL(ppo_generic_losses_kernel, dim3(blocks), dim3(PG_THREADS), 0, (hipStream_t)stream, a);
  TRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(ppo_generic_fold_kernel, dim3(1), dim3(PG_THREADS), 0, (hipStream_t)stream, workspace, blocks, A, logstd,
                     d_logstd, info);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
