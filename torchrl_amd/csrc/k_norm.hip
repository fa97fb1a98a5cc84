// K18 -- running observation normaliser of the vector env (torchrl/env/base_wrapper.py:44-121:
// update_mean_var_count, Normalizer.update_estimate / filt, NormObs.observation).
//
// State = {mean[D], var[D], count} in fp64 on the device (the reference keeps numpy float64).  One
// vector step = one batch of N observations:
//     batch_mean = mean_n x, batch_var = var_n x (population), batch_count = N          (:75-77)
//     Chan merge of (mean, var, count) with the batch                                     (:44-60)
//     out = clip((x - mean) / (sqrt(var) + 1e-4), -clip, clip)                            (:86-89)
// The batch is tiny (N x D <= a few 100 k floats), so one workgroup does moments -> merge -> filter in a
// single launch (fixed summation order: deterministic).  For env shards on several GPUs the batch
// moments {sum x, sum x^2, n} are produced separately so that ranks can all-reduce them (SUM) before every
// rank applies the same merge: all ranks then hold identical statistics, as one process would.
#include "trl_common.h"

#define NM_THREADS 1024
#define NM_MAXD 64

// per-feature sum x and sum x^2 over the N rows, fp64, fixed order: thread t owns (feature t % D, rows t / D, +R, ...)
__device__ __forceinline__ void norm_block_moments(const float* __restrict__ x, int N, int D, double* s_sum,
                                                   double* s_sq, double* s_red) {
  const int tid = threadIdx.x;
  const int R = NM_THREADS / D;                       // row lanes per feature
  const int f = tid % D, r0 = tid / D;
  double a = 0.0, b = 0.0;
  if (r0 < R)
    for (int n = r0; n < N; n += R) { const double v = (double)x[(size_t)n * D + f]; a += v; b += v * v; }
  // s_red: [2][R][D]
  if (r0 < R) { s_red[(0 * R + r0) * D + f] = a; s_red[(1 * R + r0) * D + f] = b; }
  __syncthreads();
  if (tid < D) {
    double sa = 0.0, sb = 0.0;
    for (int r = 0; r < R; ++r) { sa += s_red[(0 * R + r) * D + tid]; sb += s_red[(1 * R + r) * D + tid]; }
    s_sum[tid] = sa; s_sq[tid] = sb;
  }
  __syncthreads();
}

// Chan / Welford merge of the running state with a batch given by its moments (base_wrapper.py:44-60)
__device__ __forceinline__ void norm_merge_feature(double& mean, double& var, double count, double bsum, double bsq,
                                                   double bn) {
  const double bmean = bsum / bn;
  const double bvar = fmax(bsq / bn - bmean * bmean, 0.0);
  const double delta = bmean - mean;
  const double tot = count + bn;
  const double m2 = var * count + bvar * bn + delta * delta * count * bn / tot;
  mean = mean + delta * bn / tot;
  var = m2 / tot;
}

__device__ __forceinline__ float norm_filt(float x, double mean, double var, double clip) {
  const double z = ((double)x - mean) / (sqrt(var) + 1e-4);
  return (float)fmin(fmax(z, -clip), clip);
}

extern __shared__ double nm_lds[];

__global__ __launch_bounds__(NM_THREADS) void norm_moments_kernel(const float* __restrict__ x, int N, int D,
                                                                   double* __restrict__ sums) {
  double* s_sum = nm_lds; double* s_sq = nm_lds + NM_MAXD; double* s_red = nm_lds + 2 * NM_MAXD;
  norm_block_moments(x, N, D, s_sum, s_sq, s_red);
  if ((int)threadIdx.x < D) { sums[threadIdx.x] = s_sum[threadIdx.x]; sums[D + threadIdx.x] = s_sq[threadIdx.x]; }
  if (threadIdx.x == 0) sums[2 * D] = (double)N;
}

__global__ __launch_bounds__(256) void norm_merge_kernel(double* __restrict__ state, const double* __restrict__ sums,
                                                         int D) {
  const int f = threadIdx.x;
  const double count = state[2 * D], bn = sums[2 * D];
  if (f < D && bn > 0.0) {
    double mean = state[f], var = state[D + f];
    norm_merge_feature(mean, var, count, sums[f], sums[D + f], bn);
    state[f] = mean; state[D + f] = var;
  }
  __syncthreads();
  if (f == 0 && bn > 0.0) state[2 * D] = count + bn;
}

__global__ __launch_bounds__(256) void norm_filt_kernel(const float* __restrict__ x, const double* __restrict__ state,
                                                        float* __restrict__ out, int N, int D, float clip) {
  const int64_t total = (int64_t)N * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int f = (int)(e % D);
    out[e] = norm_filt(x[e], state[f], state[D + f], (double)clip);
  }
}

// single-process step: moments -> merge (if `update`) -> filter, one workgroup
__global__ __launch_bounds__(NM_THREADS) void norm_update_filt_kernel(const float* __restrict__ x,
                                                                       double* __restrict__ state,
                                                                       float* __restrict__ out, int N, int D, float clip,
                                                                       int update) {
  double* s_sum = nm_lds; double* s_sq = nm_lds + NM_MAXD; double* s_red = nm_lds + 2 * NM_MAXD;
  __shared__ double s_mean[NM_MAXD], s_var[NM_MAXD];
  const int tid = threadIdx.x;
  if (update) norm_block_moments(x, N, D, s_sum, s_sq, s_red);
  if (tid < D) {
    double mean = state[tid], var = state[D + tid];
    if (update) {
      norm_merge_feature(mean, var, state[2 * D], s_sum[tid], s_sq[tid], (double)N);
      state[tid] = mean; state[D + tid] = var;
    }
    s_mean[tid] = mean; s_var[tid] = var;
  }
  __syncthreads();
  if (update && tid == 0) state[2 * D] += (double)N;
  if (out) {
    const int total = N * D;
    for (int e = tid; e < total; e += NM_THREADS) { const int f = e % D; out[e] = norm_filt(x[e], s_mean[f], s_var[f], (double)clip); }
  }
}

static size_t nm_lds_bytes(int D) { return (2 * NM_MAXD + 2 * (NM_THREADS / D) * D) * sizeof(double); }

extern "C" int trl_norm_batch_moments_f64(const float* x, int N, int D, double* sums, void* stream) {
  TRL_REQUIRE(x && sums, "null pointer");
  TRL_REQUIRE(N > 0 && D > 0 && D <= NM_MAXD, "need N > 0 and 0 < D <= 64");
  hipLaunchKernelGGL(norm_moments_kernel, dim3(1), dim3(NM_THREADS), nm_lds_bytes(D), (hipStream_t)stream, x, N, D, sums);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_norm_merge_f64(double* state, const double* sums, int D, void* stream) {
  TRL_REQUIRE(state && sums, "null pointer");
  TRL_REQUIRE(D > 0 && D <= NM_MAXD, "need 0 < D <= 64");
  hipLaunchKernelGGL(norm_merge_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state, sums, D);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_norm_filt_f32(const float* x, const double* state, float* out, int N, int D, float clip,
                                 void* stream) {
  TRL_REQUIRE(N >= 0 && D > 0 && D <= NM_MAXD, "need N >= 0 and 0 < D <= 64");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(x && state && out, "null pointer");
  const int64_t blocks = ((int64_t)N * D + 255) / 256;
  hipLaunchKernelGGL(norm_filt_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, x, state, out, N, D, clip);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_norm_update_filt_f32(const float* x, double* state, float* out, int N, int D, float clip,
                                        int update, void* stream) {
  TRL_REQUIRE(x && state, "null pointer");
  TRL_REQUIRE(N > 0 && D > 0 && D <= NM_MAXD, "need N > 0 and 0 < D <= 64");
  TRL_REQUIRE((int64_t)N * D < (1ll << 31), "batch too large for the single-workgroup path");
  hipLaunchKernelGGL(norm_update_filt_kernel, dim3(1), dim3(NM_THREADS), nm_lds_bytes(D), (hipStream_t)stream, x,
                     state, out, N, D, clip, update);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
