// K10 (generic) -- fp32 MFMA GEMM family for dense layers of any shape.
//
// The fused PPO kernels cover the benchmark network; everything else that is a dense layer in the
// reference (torchrl/networks/base.py:30-44, nets.py:34-52: nn.Linear + activation; 256-wide SAC
// nets, Q nets on [obs, act], FC heads of the conv nets) goes through this family:
//     forward   Y[M,N]  = act( X[M,K] . W[N,K]^T + b[N] )          (trans_b)
//     input-grad dX[M,K] = dZ[M,N] . W[N,K]                          (plain)
//     weight-grad dW[N,K] = dZ[M,N]^T . X[M,K]   (split over M, deterministic two-pass fold)
//     dZ = dY * act'(Y)  and  db = column sums of dZ are fused into the operand loads / a side output.
// Exact fp32 on v_mfma_f32_32x32x2_f32: a workgroup of 4 waves owns a 64x64 C tile (each wave one
// 32x32 quadrant), K is walked in steps of GK = 32 through LDS tiles that are BOTH k-contiguous (A as [m][k],
// B as [n][k], row stride GK + 4 floats).  The k index of MFMA step s is (GK / 2) * (lane >> 5) + s, so a
// lane's operands of a K-step are consecutive floats: one ds_read_b128 per operand per 4 MFMAs instead of 4
// ds_read_b32 (the operand reads, not the MFMAs, paced the first version).  Arbitrary M, N, K (zero fill).
#include "trl_common.h"
#include "trl_mlp.h"

#define GM 64
#define GN 64
#define GK 32
#define LDK (GK + 4)                // LDS row stride of both operand tiles: 16-byte aligned rows of k
#define A_PER_T (GM * GK / 256)     //  8 staged A elements per thread
#define B_PER_T (GK * GN / 256)     //  8 staged B elements per thread

// act'(y) expressed through the activation OUTPUT y (tanh: 1 - y^2, relu: y > 0, none: 1)
__device__ __forceinline__ float dact_from_out(int act, float y) {
  if (act == TRL_ACT_TANH) return 1.0f - y * y;
  if (act == TRL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}

struct GemmDev {
  const float* A; const float* B; float* C;
  const float* bias;          // epilogue: + bias[n]          (nullable)
  const float* a_gate;        // operand A is A * act'(a_gate) elementwise, same shape/ld as A (nullable)
  int M, N, K, lda, ldb, ldc;
  int act;                    // epilogue activation (TRL_ACT_*), applied after bias
  int gate_act;               // activation whose derivative gates A
  int split_len;              // TA only: rows of the reduction handled by one blockIdx.z
  float* colsum;              // TA only: (splits, M) partial column sums of the gated A (nullable)
};

// TA: A is stored [Kred][M] (we need A^T); TB: B is stored [N][K] (we need B^T).
// Workgroup = 4 waves, C tile 64 x 64, one 32x32 quadrant per wave (these layers are small -- a few
// thousand rows by <= 256 columns -- so filling 256 CUs matters more than a fatter tile).  K advances
// in steps of 32; the next step's global loads are issued into registers before the current step's
// MFMAs (register double buffering).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmDev g) {
  __shared__ __attribute__((aligned(16))) float As[GM * LDK];      // [m][k]
  __shared__ __attribute__((aligned(16))) float Bs[GN * LDK];      // [n][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  int k_lo = 0, k_hi = g.K;
  float* C = g.C;
  if (TA) {                                        // split the (long) reduction dimension over blockIdx.z
    k_lo = blockIdx.z * g.split_len;
    k_hi = min(g.K, k_lo + g.split_len);
    C += (size_t)blockIdx.z * g.M * g.ldc;
  }
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
  float csum = 0.0f;                               // TA: column-sum partial of column m = tid & 63
  const int i = lane & 31, hi = lane >> 5;
  const bool want_colsum = TA && g.colsum && blockIdx.x == 0;

  float ra[A_PER_T], rb[B_PER_T];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < A_PER_T; ++t) {
      const int e = tid + 256 * t;
      int m, k;
      if (!TA) { m = e / GK; k = e - m * GK; } else { k = e / GM; m = e - k * GM; }   // coalesced along the stored-contiguous dim
      float v = 0.0f;
      if (m0 + m < g.M && k0 + k < k_hi) {
        const size_t idx = TA ? (size_t)(k0 + k) * g.lda + m0 + m : (size_t)(m0 + m) * g.lda + k0 + k;
        v = g.A[idx];
        if (g.a_gate) v *= dact_from_out(g.gate_act, g.a_gate[idx]);
      }
      ra[t] = v;
    }
#pragma unroll
    for (int t = 0; t < B_PER_T; ++t) {
      const int e = tid + 256 * t;
      int n, k;
      if (!TB) { k = e / GN; n = e - k * GN; } else { n = e / GK; k = e - n * GK; }
      float v = 0.0f;
      if (k0 + k < k_hi && n0 + n < g.N)
        v = TB ? g.B[(size_t)(n0 + n) * g.ldb + k0 + k] : g.B[(size_t)(k0 + k) * g.ldb + n0 + n];
      rb[t] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < A_PER_T; ++t) {
      const int e = tid + 256 * t;
      int m, k;
      if (!TA) { m = e / GK; k = e - m * GK; } else { k = e / GM; m = e - k * GM; }
      As[m * LDK + k] = ra[t];
      if (want_colsum) csum += ra[t];              // TA: e % GM == tid % 64 for every t -> fixed column
    }
#pragma unroll
    for (int t = 0; t < B_PER_T; ++t) {
      const int e = tid + 256 * t;
      int n, k;
      if (!TB) { k = e / GN; n = e - k * GN; } else { n = e / GK; k = e - n * GK; }
      Bs[n * LDK + k] = rb[t];
    }
  };

  if (k_lo < k_hi) fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
    stash();
    __syncthreads();
    if (k0 + GK < k_hi) fetch(k0 + GK);            // next step's loads fly under this step's MFMAs
    {
      const float* ap = As + (32 * wm + i) * LDK + (GK / 2) * hi;
      const float* bp = Bs + (32 * wn + i) * LDK + (GK / 2) * hi;
#pragma unroll
      for (int q = 0; q < GK / 8; ++q) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 4 * q);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0 = mfma32(av[r], bv[r], acc0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue ----
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + 32 * wm + rowmap(r, hi), n = n0 + 32 * wn + i;
    if (m < g.M && n < g.N) {
      float v = acc0[r];
      if (g.bias) v += g.bias[n];
      if (g.act == TRL_ACT_TANH) v = trl_tanh(v);
      else if (g.act == TRL_ACT_RELU) v = fmaxf(v, 0.0f);
      C[(size_t)m * g.ldc + n] = v;
    }
  }
  if (want_colsum) {
    // threads tid, tid+64, tid+128, tid+192 hold partials of the same column m = tid & 63
    float* s = As;                                  // reuse (all MFMA reads are behind the last barrier)
    s[tid] = csum;
    __syncthreads();
    if (tid < GM && m0 + tid < g.M)
      g.colsum[(size_t)blockIdx.z * g.M + m0 + tid] = (s[tid] + s[tid + 64]) + (s[tid + 128] + s[tid + 192]);
  }
}

// fixed-order fold of split partials: out[e] = sum_s part[s][e]
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            int n, int splits) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float a = 0.0f;
  for (int s = 0; s < splits; ++s) a += part[(size_t)s * n + e];
  out[e] = a;
}

template <bool TA, bool TB>
static int launch_gemm(const GemmDev& g, int splits, hipStream_t s) {
  dim3 grid(trl_ceil_div(g.N, GN), trl_ceil_div(g.M, GM), splits);
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB>), grid, dim3(256), 0, s, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                                  int act, void* stream) {
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  TRL_REQUIRE(x && w && y, "null pointer");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  g.A = x; g.B = w; g.C = y; g.bias = bias; g.a_gate = nullptr; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = N; g.act = act; g.gate_act = TRL_ACT_NONE; g.split_len = K; g.colsum = nullptr;
  return launch_gemm<false, true>(g, 1, (hipStream_t)stream);
}

extern "C" int trl_linear_bwd_input_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                        int M, int K, int N, void* stream) {
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  TRL_REQUIRE(dy && w && dx, "null pointer");
  GemmDev g{};
  g.A = dy; g.a_gate = y_gate; g.gate_act = gate_act; g.B = w; g.C = dx; g.bias = nullptr;
  g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = K; g.ldc = K; g.act = TRL_ACT_NONE; g.split_len = N; g.colsum = nullptr;
  return launch_gemm<false, false>(g, 1, (hipStream_t)stream);
}

extern "C" int trl_linear_bwd_weight_workspace(int M, int K, int N) {
  // floats of workspace for trl_linear_bwd_weight_f32 (split partials of dW and db)
  const int splits = M <= 0 ? 1 : (M + 255) / 256;
  return splits * (N * K + N);
}

extern "C" int trl_linear_bwd_weight_f32(const float* dy, const float* y_gate, int gate_act, const float* x, float* dw,
                                         float* db, float* workspace, int M, int K, int N, void* stream) {
  TRL_REQUIRE(M > 0 && K > 0 && N > 0, "bad sizes");
  TRL_REQUIRE(dy && x && dw && workspace, "null pointer");
  const int split_len = 256;
  const int splits = (M + split_len - 1) / split_len;
  hipStream_t s = (hipStream_t)stream;
  GemmDev g{};
  g.A = dy; g.a_gate = y_gate; g.gate_act = gate_act; g.B = x; g.C = workspace; g.bias = nullptr;
  g.M = N; g.N = K; g.K = M; g.lda = N; g.ldb = K; g.ldc = K; g.act = TRL_ACT_NONE; g.split_len = split_len;
  g.colsum = db ? workspace + (size_t)splits * N * K : nullptr;
  int rc = launch_gemm<true, false>(g, splits, s);
  if (rc) return rc;
  hipLaunchKernelGGL(fold_partials_kernel, dim3(trl_ceil_div((int64_t)N * K, 256)), dim3(256), 0, s, workspace, dw,
                     N * K, splits);
  TRL_LAUNCH_CHECK();
  if (db) {
    hipLaunchKernelGGL(fold_partials_kernel, dim3(trl_ceil_div(N, 256)), dim3(256), 0, s, g.colsum, db, N, splits);
    TRL_LAUNCH_CHECK();
  }
  return TRL_OK;
}
