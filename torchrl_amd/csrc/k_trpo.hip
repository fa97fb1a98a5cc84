// TRPO (torchrl/algo/on_policy/trpo.py:28-226) -- the element-wise pieces around the dense-layer kernels:
//   trl_trpo_surrogate_f32   L = -mean(p / (p.detach() + 1e-8) * adv) - c_ent * mean(ent) (:170-180): d L / d mean,
//                            d L / d logstd, the logged log-prob statistics and the loss value
//   trl_jvp_gate_f32         one layer of the forward-mode pass of a Fisher-vector product: out = act'(h) * (a + b)
//   trl_fisher_scale_f32     d KL / d mean at the expansion point: out = d_mu * exp(-2 logstd) / n
//   trl_ratio_loss_f32       the line search's objective -mean(exp(log pi_new - log pi_old) * adv) (:110-128)
// The Hessian of mean KL(pi_theta || pi_theta.detach()) (:62-87) of a diagonal Gaussian policy is J^T diag(1/sigma^2) J / n
// on the network parameters (second derivatives of the mean are multiplied by d KL / d mean = 0) and 2 on each logstd,
// so F v = backward(forward-mode(v) * exp(-2 logstd) / n) -- two passes over the dense-layer kernels per CG iteration.
#include "trl_common.h"
#include "trl_mlp.h"

#define TR_THREADS 256
#define TR_MAX_A 64
#define TR_SCAL 5           // lp sum, lp^2, max lp, -min lp, sum ratio * adv

__device__ __forceinline__ double tr_block_reduce(double v, bool is_max, double* smem) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = is_max ? -INFINITY : 0.0;
  for (int w = 0; w < TR_THREADS / 64; ++w) r = is_max ? fmax(r, smem[w]) : r + smem[w];
  return r;
}

struct TrpoDev {
  const float* mean; const float* logstd; const float* acts; const float* adv; float* d_mean; double* partial;
  int n, A, tanh_action; float entropy_coeff;
};

__global__ __launch_bounds__(TR_THREADS) void trpo_surrogate_kernel(TrpoDev a) {
  __shared__ double smem[TR_THREADS / 64];
  __shared__ float s_dls[TR_THREADS / 64][TR_MAX_A];
  const int b = blockIdx.x * TR_THREADS + threadIdx.x;
  const bool valid = b < a.n;
  const int A = a.A;
  const float inv_n = 1.0f / (float)a.n;
  float lp = 0.0f;
  if (valid)
    for (int o = 0; o < A; ++o) {
      const float ls = fminf(fmaxf(a.logstd[o], -20.0f), 2.0f);
      float zc;
      lp += gauss_logp_term(a.acts[(size_t)b * A + o], a.mean[(size_t)b * A + o], __expf(-2.0f * ls), ls, a.tanh_action, zc);
    }
  const float p = valid ? __expf(lp) : 0.0f;
  const float w = p / (p + 1e-8f);                                 // ratio = p / (p.detach() + 1e-8); d ratio / d log p = w
  const float adv = valid ? a.adv[b] : 0.0f;
  const float g_lp = -adv * w * inv_n;
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
      dls = pass * (g_lp * (zc * zc * ivv - 1.0f) - a.entropy_coeff * inv_n);
    }
    dls = wave_sum(dls);
    if (lane == 0) s_dls[wave][o] = dls;
  }
  double* out = a.partial + (size_t)blockIdx.x * (A + TR_SCAL);
  __syncthreads();
  if (threadIdx.x < A) {
    float s = 0.0f;
    for (int k = 0; k < TR_THREADS / 64; ++k) s += s_dls[k][threadIdx.x];
    out[threadIdx.x] = (double)s;
  }
  const double ninf = -INFINITY;
  const double vals[TR_SCAL] = {valid ? (double)lp : 0.0, valid ? (double)lp * lp : 0.0, valid ? (double)lp : ninf,
                                valid ? -(double)lp : ninf, (double)(w * adv)};
  const bool is_max[TR_SCAL] = {false, false, true, true, false};
#pragma unroll
  for (int k = 0; k < TR_SCAL; ++k) {
    const double r = tr_block_reduce(vals[k], is_max[k], smem);
    if (threadIdx.x == 0) out[A + k] = r;
  }
}

// one block: d_logstd (A); info: 0 policy loss, 1..4 log-prob mean / unbiased std / max / min
__global__ __launch_bounds__(TR_THREADS) void trpo_fold_kernel(const double* __restrict__ partial, int blocks, int A, int n,
                                                             const float* __restrict__ logstd, float entropy_coeff,
                                                             float* __restrict__ d_logstd, double* __restrict__ info) {
  __shared__ double s_out[TR_MAX_A + TR_SCAL];
  const int stride = A + TR_SCAL;
  for (int e = threadIdx.x; e < stride; e += TR_THREADS) {
    const int k = e - A;
    const bool is_max = k == 2 || k == 3;
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
    const double nn = (double)n, lp_mean = s[0] / nn;
    double ent = 0.0;                                               // Normal entropy per sample (distribution.py:78-79)
    for (int o = 0; o < A; ++o) ent += 1.4189385332046727 + fmin(fmax((double)logstd[o], -20.0), 2.0);
    info[0] = -s[4] / nn - (double)entropy_coeff * ent;
    info[1] = lp_mean; info[2] = n > 1 ? sqrt(fmax((s[1] - s[0] * lp_mean) / (nn - 1.0), 0.0)) : NAN;
    info[3] = s[2]; info[4] = -s[3];
  }
}

__global__ __launch_bounds__(TR_THREADS) void jvp_gate_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ h, int act, int64_t n,
                                                             float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * TR_THREADS + threadIdx.x;
  if (e >= n) return;
  float v = a[e] + (b ? b[e] : 0.0f);
  if (h) {
    const float y = h[e];
    v *= act == TRL_ACT_TANH ? 1.0f - y * y : (act == TRL_ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : 1.0f);
  }
  out[e] = v;
}

__global__ __launch_bounds__(TR_THREADS) void fisher_scale_kernel(const float* __restrict__ dmu, const float* __restrict__ logstd,
                                                                 int n, int A, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * TR_THREADS + threadIdx.x;
  if (e >= (int64_t)n * A) return;
  const float ls = fminf(fmaxf(logstd[e % A], -20.0f), 2.0f);
  out[e] = dmu[e] * __expf(-2.0f * ls) / (float)n;
}

__global__ __launch_bounds__(TR_THREADS) void ratio_loss_kernel(const float* __restrict__ lp_new, const float* __restrict__ lp_old,
                                                               const float* __restrict__ adv, int n, double* __restrict__ out) {
  __shared__ double smem[TR_THREADS / 64];
  double s = 0.0;
  for (int b = threadIdx.x; b < n; b += TR_THREADS) s += (double)(__expf(lp_new[b] - lp_old[b]) * adv[b]);
  s = tr_block_reduce(s, false, smem);
  if (threadIdx.x == 0) *out = -s / (double)n;
}

extern "C" int trl_trpo_surrogate_workspace(int n, int A) {
  if (n <= 0 || A <= 0 || A > TR_MAX_A) return TRL_EINVAL;
  return trl_ceil_div(n, TR_THREADS) * (A + TR_SCAL);
}

extern "C" int trl_trpo_surrogate_f32(const float* mean, const float* logstd, const float* acts, const float* adv_n, int n,
                                      int A, int tanh_action, float entropy_coeff, float* d_mean, float* d_logstd,
                                      double* info, double* workspace, void* stream) {
  TRL_REQUIRE(n > 0 && A > 0 && A <= TR_MAX_A, "bad sizes (1 <= A <= 64)");
  TRL_REQUIRE(mean && logstd && acts && adv_n && d_mean && d_logstd && info && workspace, "null pointer");
  TrpoDev a{};
  a.mean = mean; a.logstd = logstd; a.acts = acts; a.adv = adv_n; a.d_mean = d_mean; a.partial = workspace; a.n = n; a.A = A;
  a.tanh_action = tanh_action; a.entropy_coeff = entropy_coeff;
  const int blocks = trl_ceil_div(n, TR_THREADS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(trpo_surrogate_kernel, dim3(blocks), dim3(TR_THREADS), 0, s, a);
  TRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(trpo_fold_kernel, dim3(1), dim3(TR_THREADS), 0, s, workspace, blocks, A, n, logstd, entropy_coeff,
                     d_logstd, info);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_jvp_gate_f32(const float* a, const float* b, const float* h, int act, int64_t n, float* out, void* stream) {
  TRL_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(a && out, "null pointer");
  hipLaunchKernelGGL(jvp_gate_kernel, dim3(trl_ceil_div(n, TR_THREADS)), dim3(TR_THREADS), 0, (hipStream_t)stream, a, b, h, act, n, out);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_fisher_scale_f32(const float* d_mu, const float* logstd, int n, int A, float* out, void* stream) {
  TRL_REQUIRE(n > 0 && A > 0, "bad sizes");
  TRL_REQUIRE(d_mu && logstd && out, "null pointer");
  hipLaunchKernelGGL(fisher_scale_kernel, dim3(trl_ceil_div((int64_t)n * A, TR_THREADS)), dim3(TR_THREADS), 0, (hipStream_t)stream,
                     d_mu, logstd, n, A, out);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_ratio_loss_f32(const float* logp_new, const float* logp_old, const float* adv_n, int n, double* out,
                                  void* stream) {
  TRL_REQUIRE(n > 0, "empty batch");
  TRL_REQUIRE(logp_new && logp_old && adv_n && out, "null pointer");
  hipLaunchKernelGGL(ratio_loss_kernel, dim3(1), dim3(TR_THREADS), 0, (hipStream_t)stream, logp_new, logp_old, adv_n, n, out);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
