// Wave-level fp32 MFMA building blocks for the small policy/value MLPs
// (D -> H -> H -> O, activation after every hidden layer, linear head:
// torchrl/networks/base.py:8-44, nets.py:13-52).
//
// Everything is computed TRANSPOSED: a wave owns 32 samples (lane & 31) and a
// layer is  Z^T[feature][sample] = W[feature][k] * X^T[k][sample]  on
// v_mfma_f32_32x32x2_f32 (exact fp32, one rounding per product).  With that
// orientation the C/D register layout of one layer
//      reg r of lane (j, hi)  <->  feature 32*m + rowmap(r, hi), sample j
// is exactly the B-operand layout of the next layer if the k loop visits the
// features in `rowmap` order -- so activations chain through registers with no
// LDS traffic; only the weights (A operand) are fetched from LDS, one
// ds_read_b32 per 64-cycle MFMA.  Weight-gradient GEMMs contract over samples
// and need the other orientation (lane = feature); that is a 32x33-padded LDS
// transpose per 32x32 tile (conflict free both ways).
#pragma once
#include "trl_common.h"

__device__ __forceinline__ constexpr int rowmap(int r, int hi) { return (r & 3) + ((r >> 2) << 3) + (hi << 2); }

// number of k-steps (pairs of input features) needed to cover D <= 32 inputs in rowmap order
__host__ __device__ constexpr int ksteps_for(int d) {
  int n = 0;
  for (int r = 0; r < 16; ++r) if (((r & 3) + ((r >> 2) << 3)) < d) n = r + 1;
  return n;
}

__host__ __device__ constexpr int align4(int x) { return (x + 3) & ~3; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// tanh(x) = 1 - 2 / (2^(x * 2 log2 e) + 1): one multiply, v_exp_f32, one add, v_rcp_f32, one fma -- valid for both signs,
// saturates cleanly (2^inf -> rcp 0 -> 1; 2^-inf -> rcp 1 -> -1).  Absolute error < 2e-7 over the real line
// (tests/test_kernels_gpu.py checks the dense-layer kernels built on it against float64 tanh); near zero the error
// is absolute, not relative -- |x| <= 1e-3 costs up to 1e-4 of tanh(x) itself, invisible next to fp32 GEMM round-off.
// A vector instruction takes 4 cycles per wave on gfx950 and a transcendental 16 (tools/time_grad.py phase clocks: 63 ticks
// per tanh with the former 12-instruction form that also carried an x - x^3/3 branch for |x| < 0.04), and fp32 MFMA does
// not overlap with vector work (tools/ubench/mfma_valu.hip), so every instruction here is wall-clock time in the kernels.
__device__ __forceinline__ float trl_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// One action dimension of log pi(a|s) for a (Tanh)Normal policy, the reference's formula
// (torchrl/policies/distribution.py:33-45): atanh(a) = log((1+a)/(1-a))/2, Normal log-density
// minus log(1 - a^2 + 1e-6).  Shared by the collector, the PPO loss and trl_gauss_logp_f32 so
// that log pi and log pi_old of the same (s, a, params) are bit-identical (ratio == 1).
// Both logarithms are v_log_f32 (log2, ~1 ulp; arguments are normal numbers: (1+a)/(1-a) in [3e-8, 7e7], 1 - a^2 + 1e-6
// >= 1e-6) times ln 2: rel. error ~1e-7 of each log term.  (`__logf` expands to a 15-instruction denormal-safe,
// extended-precision sequence here -- 2.5 k ticks per 16-sample policy tile for the four terms.)  zc = atanh(a) - mean.
__device__ __forceinline__ float gauss_logp_term(float act, float mean, float inv_var, float ls, int tanh_action,
                                                 float& zc) {
  float pre = act, corr = 0.0f;
  if (tanh_action) {
    pre = 0.34657359027997264f * __builtin_amdgcn_logf((1.0f + act) * __builtin_amdgcn_rcpf(1.0f - act));   // ln 2 / 2
    corr = 0.6931471805599453f * __builtin_amdgcn_logf(fmaf(-act, act, 1.0f) + 1e-6f);
  }
  zc = pre - mean;
  return -(zc * zc) * 0.5f * inv_var - ls - 0.91893853320467274f - corr;
}

template <int ACT> __device__ __forceinline__ float act_fn(float z) {
  if (ACT == TRL_ACT_TANH) return trl_tanh(z);
  return fmaxf(z, 0.0f);
}
// derivative expressed through the activation output h
template <int ACT> __device__ __forceinline__ float act_grad(float h) {
  if (ACT == TRL_ACT_TANH) return 1.0f - h * h;
  return h > 0.0f ? 1.0f : 0.0f;
}
template <int ACT> __device__ __forceinline__ f32x16 act_tile(f32x16 z) {
  f32x16 h;
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = act_fn<ACT>(z[r]);
  return h;
}

// ---- flat parameter block offsets (global memory, see include/trl_hip.h) ----
template <int D, int H, int O> struct MlpFlat {
  static constexpr int W1 = 0, B1 = H * D, W2 = B1 + H, B2 = W2 + H * H, W3 = B2 + H, B3 = W3 + O * H,
                       LS = B3 + O;                 // logstd (policies only)
  static constexpr int P_VF = LS, P_PF = LS + O;
};

// ---- LDS copy of one parameter block, strides padded to be bank-conflict free ----
template <int D, int H, int O> struct MlpLds {
  static_assert(D <= 32, "input dim > 32 not instantiated");
  static_assert(H % 32 == 0, "hidden width must be a multiple of 32");
  static constexpr int LD1 = (D % 2 == 0) ? D + 1 : D;      // odd -> lanes i*LD1 hit distinct banks (b32 reads)
  static constexpr int LD2 = H + 4;                          // 16-B aligned rows, ds_read_b128 conflict free
  static constexpr int W1 = 0;
  static constexpr int B1 = align4(W1 + H * LD1);
  static constexpr int W2 = B1 + H;
  static constexpr int B2 = align4(W2 + H * LD2);
  static constexpr int W3 = B2 + H;                           // [O][H], rows 16-byte aligned
  static constexpr int B3 = W3 + O * H;
  static constexpr int LS = B3 + align4(O);
  static constexpr int SIZE = LS + align4(O);

  // cooperative global -> LDS copy by `nthreads` threads
  __device__ static void load(float* sp, const float* __restrict__ gp, bool has_logstd, int tid, int nthreads) {
    using F = MlpFlat<D, H, O>;
    for (int e = tid; e < H * D; e += nthreads) sp[W1 + (e / D) * LD1 + (e % D)] = gp[F::W1 + e];
    for (int e = tid; e < H * H; e += nthreads) sp[W2 + (e / H) * LD2 + (e % H)] = gp[F::W2 + e];
    for (int e = tid; e < O * H; e += nthreads) sp[W3 + e] = gp[F::W3 + e];
    for (int e = tid; e < H; e += nthreads) { sp[B1 + e] = gp[F::B1 + e]; sp[B2 + e] = gp[F::B2 + e]; }
    for (int e = tid; e < align4(O); e += nthreads) {
      sp[B3 + e] = e < O ? gp[F::B3 + e] : 0.0f;
      sp[LS + e] = (has_logstd && e < O) ? gp[F::LS + e] : 0.0f;
    }
  }
};

// accumulator tile initialised with the bias of features [base, base+32)
__device__ __forceinline__ f32x16 bias_tile(const float* b, int hi) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(b + 8 * q + 4 * hi);
    acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
  }
  return acc;
}

__device__ __forceinline__ f32x16 zero_tile() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  return z;
}

// first layer, output feature tile `mo`: acc += W1[32mo + i][k] * x^T[k][j], k in rowmap order.
// xb[s] = x[sample j][rowmap(s, hi)] (0 where rowmap >= D)
template <int D, int LD1, int KS>
__device__ __forceinline__ f32x16 layer1_tile(f32x16 acc, const float* w1, int mo, const float (&xb)[KS], int i, int hi) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rowmap(s, hi);
    const float a = (k < D) ? w1[(32 * mo + i) * LD1 + k] : 0.0f;
    acc = mfma32(a, xb[s], acc);
  }
  return acc;
}

// hidden layer, output tile `mo`, all NT source tiles in registers:
//   acc += W[32mo + i][32m + rowmap(r, hi)] * h[m][r]
// rowmap(4q..4q+3, hi) are 4 consecutive features, so one 16-byte LDS read feeds 4 MFMAs
template <int LD>
__device__ __forceinline__ f32x16 layer_tile_1src(f32x16 acc, const float* w, int mo, int m, const f32x16& h, int i, int hi) {
  static_assert(LD % 4 == 0, "rows must be 16-byte aligned");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + (32 * mo + i) * LD + 32 * m + 8 * q + 4 * hi);
    acc = mfma32(a[0], h[4 * q + 0], acc); acc = mfma32(a[1], h[4 * q + 1], acc);
    acc = mfma32(a[2], h[4 * q + 2], acc); acc = mfma32(a[3], h[4 * q + 3], acc);
  }
  return acc;
}
template <int NT, int LD>
__device__ __forceinline__ f32x16 layer_tile(f32x16 acc, const float* w, int mo, const f32x16 (&h)[NT], int i, int hi) {
#pragma unroll
  for (int m = 0; m < NT; ++m) acc = layer_tile_1src<LD>(acc, w, mo, m, h[m], i, hi);
  return acc;
}

// same with the weight matrix read transposed (backward: dH_in^T = W^T * dZ_out^T):
//   acc += W[32m + rowmap(r, hi)][32mo + i] * dz[m][r]
template <int LD>
__device__ __forceinline__ f32x16 layer_tile_wT_1src(f32x16 acc, const float* w, int mo, int m, const f32x16& dz, int i, int hi) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    acc = mfma32(w[(32 * m + rowmap(r, hi)) * LD + 32 * mo + i], dz[r], acc);
  return acc;
}
template <int NT, int LD>
__device__ __forceinline__ f32x16 layer_tile_wT(f32x16 acc, const float* w, int mo, const f32x16 (&dz)[NT], int i, int hi) {
#pragma unroll
  for (int m = 0; m < NT; ++m) acc = layer_tile_wT_1src<LD>(acc, w, mo, m, dz[m], i, hi);
  return acc;
}

// linear head on the VALU: out[o] = b3[o] + sum_f W3[o][f] h[f]; lane (j, hi) holds half the
// features of sample j, the other half sits in lane j + 32.
template <int NT, int H, int O>
__device__ __forceinline__ void head_fwd(const float* w3, const float* b3, const f32x16 (&h)[NT], int hi, float (&out)[O]) {
#pragma unroll
  for (int o = 0; o < O; ++o) {
    float p = 0.0f;
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3 + o * H + 32 * m + 8 * q + 4 * hi);
        p = fmaf(w[0], h[m][4 * q + 0], p); p = fmaf(w[1], h[m][4 * q + 1], p);
        p = fmaf(w[2], h[m][4 * q + 2], p); p = fmaf(w[3], h[m][4 * q + 3], p);
      }
    out[o] = p + __shfl_xor(p, 32, 64) + b3[o];
  }
}

// dH^T[f][j] = sum_o W3[o][f] dout[o]  (lane-local)
template <int NT, int H, int O>
__device__ __forceinline__ void head_bwd(const float* w3, const float (&dout)[O], int hi, f32x16 (&dh)[NT]) {
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    dh[m] = zero_tile();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3 + o * H + 32 * m + 8 * q + 4 * hi);
        dh[m][4 * q + 0] = fmaf(w[0], dout[o], dh[m][4 * q + 0]);
        dh[m][4 * q + 1] = fmaf(w[1], dout[o], dh[m][4 * q + 1]);
        dh[m][4 * q + 2] = fmaf(w[2], dout[o], dh[m][4 * q + 2]);
        dh[m][4 * q + 3] = fmaf(w[3], dout[o], dh[m][4 * q + 3]);
      }
  }
}

// ---- 32x32 tile transpose through a wave-private LDS scratch T[feature][33] ----
#define TRL_TLD 33
template <int NT>
__device__ __forceinline__ void tile_store_T(float* T, const f32x16 (&h)[NT], int j, int hi) {
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(32 * m + rowmap(r, hi)) * TRL_TLD + j] = h[m][r];
}
// one tile: T[(32m + feature)][sample]
__device__ __forceinline__ void tile_store_T1(float* T, int m, const f32x16& h, int j, int hi) {
#pragma unroll
  for (int r = 0; r < 16; ++r) T[(32 * m + rowmap(r, hi)) * TRL_TLD + j] = h[r];
}
// read a tile back in the layout it was stored from (lane = sample j, reg r = feature rowmap(r, hi))
__device__ __forceinline__ f32x16 tile_load_T1(const float* T, int m, int j, int hi) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = T[(32 * m + rowmap(r, hi)) * TRL_TLD + j];
  return v;
}
// read back with lane = feature i of tile m, reg r = sample rowmap(r, hi)
__device__ __forceinline__ f32x16 tile_load_N(const float* T, int m, int i, int hi) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = T[(32 * m + i) * TRL_TLD + rowmap(r, hi)];
  return v;
}

// ------------------------------------------------------------------------------------------------
// 16x16x4 variant (v_mfma_f32_16x16x4_f32): a wave owns 16 features x 16 samples; lane (j = lane & 15,
// g = lane >> 4).  A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], C/D reg r <-> row 4g + r.
#define TL 17                                   // LDS staging row stride (16 samples + 1)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// T-layout tile (MFMA C/D): reg r of lane (j, g) <-> feature 16*slice + 4g + r, sample j.
// Staging rows are permuted inside a slice -- feature 4g + r sits in row 4r + g -- so that the two
// lane groups of a 32-lane half write rows one apart (17 banks apart) instead of 4 rows apart
// (4*17 = 68 = 4 banks apart, a 2-way conflict); srow() gives the row of feature f for the
// lane-equals-feature reads, which stay conflict free because the stride is odd.
__device__ __forceinline__ constexpr int srow(int f) { return ((f & 3) << 2) | (f >> 2); }
__device__ __forceinline__ void store_T(float* S, int slice, const f32x4& t, int j, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(16 * slice + 4 * r + g) * TL + j] = t[r];
}
__device__ __forceinline__ f32x4 load_T(const float* S, int slice, int j, int g) {
  f32x4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = S[(16 * slice + 4 * r + g) * TL + j];
  return t;
}

