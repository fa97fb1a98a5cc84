// V-MPO (torchrl/algo/on_policy/v_mpo.py:57-181) -- the loss half of VMPO.update as stand-alone kernels; the
// networks' layers run on the dense-layer GEMM family (k_gemm.hip), the host picks the top half of the minibatch by
// normalised advantage (v_mpo.py:64-70).
//   trl_adv_normalize_f32     adv_n = (adv - mean) / (std_unbiased + eps) from the minibatch statistics (:175-177)
//   trl_mse_value_loss_f32    MSE(V, R): loss sum and d/dV (:136-153)
//   trl_vmpo_losses_f32       on the selected samples: phi = softmax(adv_n / eta), log pi (TanhNormal, the PPO helper),
//                             KL(pi || pi_target) of the diagonal Gaussians, L_pi = mean(-phi log pi + alpha KL);
//                             d L_pi / d mean, d L_pi / d logstd, the logged statistics, the gradients of the two dual
//                             variables (L_eta = eta eps + eta log mean exp(adv_n / eta), L_alpha = alpha eps - alpha
//                             mean KL) and their Adam(eps 1e-5) step + clamp at 1e-8 (:72-117)
// Partials are folded in fixed order (deterministic).
#include "trl_common.h"
#include "trl_mlp.h"

#define VM_THREADS 256
#define VM_MAX_A 64
#define VM_SCAL 9           // lp sum, lp^2, max lp, -min lp, kl sum, kl^2, max kl, -min kl, policy-loss sum

__device__ __forceinline__ double vm_block_reduce(double v, bool is_max, double* smem) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = is_max ? -INFINITY : 0.0;
  for (int w = 0; w < VM_THREADS / 64; ++w) r = is_max ? fmax(r, smem[w]) : r + smem[w];
  return r;
}

__global__ __launch_bounds__(VM_THREADS) void adv_normalize_kernel(const float* __restrict__ advs,
                                                                  const double* __restrict__ raw, double ng, int B,
                                                                  float eps, float* __restrict__ out) {
  const int b = blockIdx.x * VM_THREADS + threadIdx.x;
  if (b >= B) return;
  const double mean = raw[0] / ng, var = (raw[1] - raw[0] * raw[0] / ng) / (ng - 1.0);
  out[b] = (advs[b] - (float)mean) * (1.0f / ((float)sqrt(fmax(var, 0.0)) + eps));
}

__global__ __launch_bounds__(VM_THREADS) void mse_value_kernel(const float* __restrict__ v, const float* __restrict__ rets,
                                                              int B, float inv_n, float* __restrict__ d_v,
                                                              double* __restrict__ loss_sum) {
  __shared__ double smem[VM_THREADS / 64];
  double l = 0.0;
  for (int b = threadIdx.x; b < B; b += VM_THREADS) {
    const float d = v[b] - rets[b];
    d_v[b] = 2.0f * d * inv_n;
    l += (double)(d * d);
  }
  l = vm_block_reduce(l, false, smem);
  if (threadIdx.x == 0) *loss_sum = l;
}

// softmax normaliser over the selected advantages: ws[0] = max(a / eta), ws[1] = sum exp(a / eta - max),
// ws[2] = sum exp(a / eta - max) * a
__global__ __launch_bounds__(VM_THREADS) void vmpo_softmax_kernel(const float* __restrict__ adv, int n,
                                                                 const float* __restrict__ dual, double* __restrict__ ws) {
  __shared__ double smem[VM_THREADS / 64];
  const float inv_eta = 1.0f / dual[0];
  double mx = -INFINITY;
  for (int b = threadIdx.x; b < n; b += VM_THREADS) mx = fmax(mx, (double)(adv[b] * inv_eta));
  mx = vm_block_reduce(mx, true, smem);
  double s = 0.0, t = 0.0;
  for (int b = threadIdx.x; b < n; b += VM_THREADS) {
    const double e = exp((double)(adv[b] * inv_eta) - mx);
    s += e; t += e * (double)adv[b];
  }
  s = vm_block_reduce(s, false, smem);
  t = vm_block_reduce(t, false, smem);
  if (threadIdx.x == 0) { ws[0] = mx; ws[1] = s; ws[2] = t; }
}

struct VmpoDev {
  const float* mean; const float* tmean; const float* logstd; const float* tlogstd; const float* acts; const float* adv;
  const float* dual; const double* soft; float* d_mean; double* partial;
  int n, A, tanh_action;
};

__global__ __launch_bounds__(VM_THREADS) void vmpo_losses_kernel(VmpoDev a) {
  __shared__ double smem[VM_THREADS / 64];
  __shared__ float s_dls[VM_THREADS / 64][VM_MAX_A];
  const int b = blockIdx.x * VM_THREADS + threadIdx.x;
  const bool valid = b < a.n;
  const int A = a.A;
  const float eta = a.dual[0], alpha = a.dual[1];
  const float inv_n = 1.0f / (float)a.n;
  const float phi = valid ? (float)(exp((double)(a.adv[b] / eta) - a.soft[0]) / a.soft[1]) : 0.0f;
  const float g_lp = -phi * inv_n;                                // d L_pi / d log pi_b
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lp = 0.0f, kl = 0.0f;
  for (int o = 0; o < A; ++o) {
    float dls = 0.0f;
    if (valid) {
      const float raw = a.logstd[o], traw = a.tlogstd[o];
      const float ls = fminf(fmaxf(raw, -20.0f), 2.0f), tls = fminf(fmaxf(traw, -20.0f), 2.0f);
      const float pass = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
      const float ivv = __expf(-2.0f * ls), tivv = __expf(-2.0f * tls);
      const float m = a.mean[(size_t)b * A + o], mt = a.tmean[(size_t)b * A + o];
      float zc;
      lp += gauss_logp_term(a.acts[(size_t)b * A + o], m, ivv, ls, a.tanh_action, zc);
      // KL(N(m, s) || N(mt, st)) = log(st / s) + (s^2 + (m - mt)^2) / (2 st^2) - 1/2
      const float var = __expf(2.0f * ls), dm = m - mt;
      kl += (tls - ls) + 0.5f * (var + dm * dm) * tivv - 0.5f;
      a.d_mean[(size_t)b * A + o] = g_lp * zc * ivv + alpha * inv_n * dm * tivv;
      dls = pass * (g_lp * (zc * zc * ivv - 1.0f) + alpha * inv_n * (var * tivv - 1.0f));
    }
    dls = wave_sum(dls);
    if (lane == 0) s_dls[wave][o] = dls;
  }
  double* out = a.partial + (size_t)blockIdx.x * (A + VM_SCAL);
  __syncthreads();
  if (threadIdx.x < A) {
    float s = 0.0f;
    for (int w = 0; w < VM_THREADS / 64; ++w) s += s_dls[w][threadIdx.x];
    out[threadIdx.x] = (double)s;
  }
  const double ninf = -INFINITY;
  const double pl = valid ? (double)(-phi * lp + alpha * kl) : 0.0;
  const double vals[VM_SCAL] = {valid ? (double)lp : 0.0, valid ? (double)lp * lp : 0.0, valid ? (double)lp : ninf,
                                valid ? -(double)lp : ninf, valid ? (double)kl : 0.0, valid ? (double)kl * kl : 0.0,
                                valid ? (double)kl : ninf, valid ? -(double)kl : ninf, pl};
  const bool is_max[VM_SCAL] = {false, false, true, true, false, false, true, true, false};
#pragma unroll
  for (int k = 0; k < VM_SCAL; ++k) {
    const double r = vm_block_reduce(vals[k], is_max[k], smem);
    if (threadIdx.x == 0) out[A + k] = r;
  }
}

// one block: fold; d_logstd; info (see include/trl_hip.h); dual gradients, their Adam step and clamp
__global__ __launch_bounds__(VM_THREADS) void vmpo_fold_kernel(const double* __restrict__ partial, int blocks, int A, int n,
                                                             const double* __restrict__ soft, float* __restrict__ dual,
                                                             float eta_eps, float alpha_eps, float lr,
                                                             float* __restrict__ d_logstd, double* __restrict__ info) {
  __shared__ double s_out[VM_MAX_A + VM_SCAL];
  const int stride = A + VM_SCAL;
  for (int e = threadIdx.x; e < stride; e += VM_THREADS) {
    const int k = e - A;
    const bool is_max = k == 2 || k == 3 || k == 6 || k == 7;
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
    const double nn = (double)n;
    const double lp_mean = s[0] / nn, kl_mean = s[4] / nn;
    info[0] = s[8] / nn;                                            // Training/policy_loss
    info[1] = lp_mean; info[2] = n > 1 ? sqrt(fmax((s[1] - s[0] * lp_mean) / (nn - 1.0), 0.0)) : NAN;
    info[3] = s[2]; info[4] = -s[3];
    info[5] = kl_mean; info[6] = n > 1 ? sqrt(fmax((s[5] - s[4] * kl_mean) / (nn - 1.0), 0.0)) : NAN;
    info[7] = s[6]; info[8] = -s[7];
    const double eta = (double)dual[0], alpha = (double)dual[1];
    info[9] = alpha * alpha_eps - alpha * kl_mean;                  // Training/alpha_loss
    // d/d eta [eta eps + eta log mean exp(a / eta)] = eps + log mean exp(a / eta) - (sum_b softmax_b a_b) / eta
    const double g_eta = (double)eta_eps + (log(soft[1] / nn) + soft[0]) - soft[2] / (soft[1] * eta);
    const double g_alpha = (double)alpha_eps - kl_mean;
    // Adam(lr, betas (0.9, 0.999), eps 1e-5) on (eta, alpha): dual[2..3] exp_avg, dual[4..5] exp_avg_sq, dual[6] steps
    const float t = dual[6] + 1.0f;
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    const double g[2] = {g_eta, g_alpha};
    for (int k = 0; k < 2; ++k) {
      const float gk = (float)g[k];
      const float m = 0.9f * dual[2 + k] + 0.1f * gk;
      const float v = 0.999f * dual[4 + k] + 0.001f * gk * gk;
      dual[2 + k] = m; dual[4 + k] = v;
      const float denom = sqrtf(v) / (float)sqrt(bc2) + 1e-5f;
      dual[k] = fmaxf(dual[k] - (lr / (float)bc1) * (m / denom), 1e-8f);
    }
    dual[6] = t;
    info[10] = (double)dual[1];                                     // Training/alpha (after the step)
    info[11] = (double)dual[0];                                     // Training/eta
  }
}

extern "C" int trl_adv_normalize_f32(const float* advs, const double* adv_raw, double n_global, int B, float eps,
                                     float* out, void* stream) {
  TRL_REQUIRE(B > 0 && n_global >= 2.0, "need at least two samples");
  TRL_REQUIRE(advs && adv_raw && out, "null pointer");
  hipLaunchKernelGGL(adv_normalize_kernel, dim3(trl_ceil_div(B, VM_THREADS)), dim3(VM_THREADS), 0, (hipStream_t)stream, advs,
                     adv_raw, n_global, B, eps, out);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_mse_value_loss_f32(const float* v, const float* rets, int B, double n_global, float* d_v,
                                      double* loss_sum, void* stream) {
  TRL_REQUIRE(B > 0 && n_global >= 1.0, "bad sizes");
  TRL_REQUIRE(v && rets && d_v && loss_sum, "null pointer");
  hipLaunchKernelGGL(mse_value_kernel, dim3(1), dim3(VM_THREADS), 0, (hipStream_t)stream, v, rets, B, (float)(1.0 / n_global),
                     d_v, loss_sum);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_vmpo_losses_workspace(int n, int A) {
  if (n <= 0 || A <= 0 || A > VM_MAX_A) return TRL_EINVAL;
  return 4 + trl_ceil_div(n, VM_THREADS) * (A + VM_SCAL);          // doubles
}

extern "C" int trl_vmpo_losses_f32(const float* mean, const float* target_mean, const float* logstd,
                                   const float* target_logstd, const float* acts, const float* adv_n, float* dual_state,
                                   int n, int A, int tanh_action, float eta_eps, float alpha_eps, float dual_lr,
                                   float* d_mean, float* d_logstd, double* info, double* workspace, void* stream) {
  TRL_REQUIRE(n > 0 && A > 0 && A <= VM_MAX_A, "bad sizes (1 <= A <= 64)");
  TRL_REQUIRE(mean && target_mean && logstd && target_logstd && acts && adv_n && dual_state && d_mean && d_logstd && info &&
              workspace, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(vmpo_softmax_kernel, dim3(1), dim3(VM_THREADS), 0, s, adv_n, n, dual_state, workspace);
  TRL_LAUNCH_CHECK();
  VmpoDev a{};
  a.mean = mean; a.tmean = target_mean; a.logstd = logstd; a.tlogstd = target_logstd; a.acts = acts; a.adv = adv_n;
  a.dual = dual_state; a.soft = workspace; a.d_mean = d_mean; a.partial = workspace + 4; a.n = n; a.A = A;
  a.tanh_action = tanh_action;
  const int blocks = trl_ceil_div(n, VM_THREADS);
  hipLaunchKernelGGL(vmpo_losses_kernel, dim3(blocks), dim3(VM_THREADS), 0, s, a);
  TRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(vmpo_fold_kernel, dim3(1), dim3(VM_THREADS), 0, s, workspace + 4, blocks, A, n, workspace, dual_state,
                     eta_eps, alpha_eps, dual_lr, d_logstd, info);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
