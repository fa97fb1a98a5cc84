// Shared by k_gemm.hip (implicit-GEMM conv modes) and k_conv1.hip (direct first-layer kernels): the virtual im2col
// matrix of a conv window over the input, as address arithmetic.
#pragma once
#include "trl_common.h"

// Implicit-GEMM operand (first conv layer of CNNBase, networks/base.py:59-107, on the replay buffer's uint8 NCHW
// frame stacks): row m = (b, oy, ox), reduction index k = (c, i, j) in nn.Conv2d's weight order, element
//     cols[m][k] = frames[b][c][oy * sh + i][ox * sw + j] * scale + shift          (ScaledFloatFrame on the fly)
// With kw, sw and W multiples of 4 the 4 consecutive k of a slot are 4 consecutive, 4-byte aligned bytes: one
// dword load per slot, converted when the panel is written to LDS.  The 210 MB im2col buffer of cfg 5 never exists.
// The later conv layers read fp32 channels-last activations (B, H, W, C): with the reduction index ordered
// k' = (i, j, c) a window row is ONE contiguous run of kw * C floats, so a slot is one 16-byte load (C % 4 == 0);
// the nn.Conv2d weight stays in its (Cout, C, kh, kw) layout and is gathered with the matching permutation
// (it is a few thousand floats), and the weight gradient is un-permuted by the fold.
struct ConvSrc {
  const uint8_t* frames;      // uint8 NCHW source (CONV 1 / 2)
  const float* x;             // fp32 NHWC source (CONV 3 / 4)
  int C, H, W, kh, kw, sh, sw, Ho, Wo;
  float scale, shift;
  uint32_t hw_magic, hw_shift, w_magic, w_shift;    // division by Ho * Wo and by Wo (multiply-high form)
};

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t d, uint32_t magic, uint32_t shift) {
  if (d == 1) return n;                             // uniform
  const uint32_t q = __umulhi(n, magic);
  return (((n - q) >> 1) + q) >> shift;
}
// byte offset of window (b, oy, ox), channel 0, tap (0, 0)
__device__ __forceinline__ uint32_t conv_row_offset(const ConvSrc& cv, uint32_t m) {
  const uint32_t hw = (uint32_t)(cv.Ho * cv.Wo);
  const uint32_t b = fastdiv(m, hw, cv.hw_magic, cv.hw_shift), p = m - b * hw;
  const uint32_t oy = fastdiv(p, (uint32_t)cv.Wo, cv.w_magic, cv.w_shift), ox = p - oy * cv.Wo;
  return ((b * cv.C) * cv.H + oy * cv.sh) * cv.W + ox * cv.sw;
}
// byte offset of reduction index k = (c, i, j) relative to the window origin
__device__ __forceinline__ uint32_t conv_tap_offset(const ConvSrc& cv, uint32_t k) {
  const uint32_t khw = (uint32_t)(cv.kh * cv.kw);
  const uint32_t c = k / khw, rem = k - c * khw, i = rem / (uint32_t)cv.kw, j = rem - i * cv.kw;
  return (c * cv.H + i) * cv.W + j;
}
// fp32 NHWC: float offset of window (b, oy, ox), and of reduction index k' = (i, j, c) inside the window
__device__ __forceinline__ uint32_t nhwc_row_offset(const ConvSrc& cv, uint32_t m) {
  const uint32_t hw = (uint32_t)(cv.Ho * cv.Wo);
  const uint32_t b = fastdiv(m, hw, cv.hw_magic, cv.hw_shift), p = m - b * hw;
  const uint32_t oy = fastdiv(p, (uint32_t)cv.Wo, cv.w_magic, cv.w_shift), ox = p - oy * cv.Wo;
  return ((b * cv.H + oy * cv.sh) * cv.W + ox * cv.sw) * cv.C;
}
__device__ __forceinline__ uint32_t nhwc_tap_offset(const ConvSrc& cv, uint32_t k) {
  const uint32_t run = (uint32_t)(cv.kw * cv.C);
  const uint32_t i = k / run;
  return i * (uint32_t)(cv.W * cv.C) + (k - i * run);
}
__device__ __forceinline__ f32x4 conv_unpack(uint32_t u, float scale, float shift) {
  f32x4 v = {fmaf((float)(u & 0xffu), scale, shift), fmaf((float)((u >> 8) & 0xffu), scale, shift),
             fmaf((float)((u >> 16) & 0xffu), scale, shift), fmaf((float)(u >> 24), scale, shift)};
  return v;
}

// multiply-high constants for fastdiv (round-up method; d >= 2, d == 1 is handled by the caller's branch)
static inline void fastdiv_gen(uint32_t d, uint32_t& magic, uint32_t& shift) {
  if (d <= 1) { magic = 0; shift = 0; return; }
  const uint32_t L = 31 - (uint32_t)__builtin_clz(d);
  if ((d & (d - 1)) == 0) { magic = 0; shift = L - 1; return; }
  const uint64_t num = (uint64_t)1 << (32 + L);
  uint32_t m = (uint32_t)(num / d);
  const uint32_t rem = (uint32_t)(num - (uint64_t)m * d);
  m += m;
  const uint32_t twice = rem + rem;
  if (twice >= d || twice < rem) m += 1;
  magic = m + 1; shift = L;
}

// geometry of an implicit transposed convolution (k_conv_dx.hip) and the re-ordering of its weights:
// wprep[class block][tap][co / 16][r][cb][gq][j] = W[co = 16 chunk + 4 gq + r][c = 16 cb + j][i][j_tap]: MFMA step
// (chunk, r) of column block cb reads 64 consecutive floats.  Class blocks follow each other in class order.
struct DxGeom {
  int B, Cin, H, W, kh, kw, sh, sw, Ho, Wo, Cout;
  int gate_act, x_gate_act, tiles_per_wg;
  int dbg;                      // TRL_EXP_DX builds only (tools/bench_convdx.py): phases to skip
};
__host__ __device__ inline int dx_class_taps(const DxGeom& g, int py, int px) {
  const int nti = py < g.kh ? (g.kh - py + g.sh - 1) / g.sh : 0, ntj = px < g.kw ? (g.kw - px + g.sw - 1) / g.sw : 0;
  return nti * ntj;
}
__device__ __forceinline__ void dx_prep_range(const float* __restrict__ w, float* __restrict__ wprep, const DxGeom& g, int first,
                                              int stride) {
  const int total = g.Cout * g.Cin * g.kh * g.kw, CB = g.Cin >> 4, tap_floats = g.Cout * g.Cin;
  for (int e = first; e < total; e += stride) {
    const int jt = e % g.kw, it = (e / g.kw) % g.kh, c = (e / (g.kw * g.kh)) % g.Cin, co = e / (g.kw * g.kh * g.Cin);
    const int py = it % g.sh, px = jt % g.sw, ti = it / g.sh, tj = jt / g.sw;
    int off = 0;
    for (int cls = 0; cls < py * g.sw + px; ++cls) off += dx_class_taps(g, cls / g.sw, cls % g.sw) * tap_floats;
    const int ntj = (g.kw - px + g.sw - 1) / g.sw;
    const int chunk = co >> 4, gq = (co >> 2) & 3, r = co & 3, cb = c >> 4, j = c & 15;
    wprep[off + (ti * ntj + tj) * tap_floats + (((chunk * 4 + r) * CB + cb) * 4 + gq) * 16 + j] = w[e];
  }
}
// Jobs that ride on a first-layer forward launch (extra workgroups): the LATER conv layers' weights (Cout, C, kh * kw)
// copied into the (i, j, c) reduction order of the channels-last implicit GEMM, dst[n][ij * C + c] = src[n][c * khw + ij]
// (its B operand is then dense 16-byte loads instead of four strided 4-byte loads per slot), and the re-ordered weights
// of the backward pass's implicit transposed convolutions.
#define CONV_PERM_MAX 4
struct PermJobs {
  int n; const float* src[CONV_PERM_MAX]; float* dst[CONV_PERM_MAX]; int cout[CONV_PERM_MAX], c[CONV_PERM_MAX], khw[CONV_PERM_MAX];
  int n_dx; const float* dx_w[CONV_PERM_MAX]; float* dx_ws[CONV_PERM_MAX]; DxGeom dx_g[CONV_PERM_MAX];
};
__device__ __forceinline__ void conv_perm_jobs(const PermJobs& pj, int block, int n_blocks, int tid, int n_threads) {
  for (int k = 0; k < pj.n; ++k) {
    const int C = pj.c[k], khw = pj.khw[k], K = C * khw, total = pj.cout[k] * K;
    for (int e = block * n_threads + tid; e < total; e += n_blocks * n_threads) {
      const int row = e / K, kp = e - row * K, ij = kp / C, c = kp - ij * C;   // e = destination index
      pj.dst[k][e] = pj.src[k][row * K + c * khw + ij];
    }
  }
  for (int k = 0; k < pj.n_dx; ++k) dx_prep_range(pj.dx_w[k], pj.dx_ws[k], pj.dx_g[k], block * n_threads + tid, n_blocks * n_threads);
}
#define CONV_PERM_BLOCKS 64

// ---- direct kernels for a narrow first layer (k_conv1.hip); used by the trl_conv_*_u8 entry points when they apply ----
bool trl_conv1_direct_ok(int K, int Cout, const float* w);
// (frames2 .. y2: a second problem of the same geometry in the same launch, or nulls)
int trl_conv1_direct_fwd(const ConvSrc& cv, const float* w, const float* bias, float* y, int M, int K, int Cout, int act,
                         const PermJobs& pj, hipStream_t stream, const uint8_t* frames2 = nullptr, const float* w2 = nullptr,
                         const float* bias2 = nullptr, float* y2 = nullptr);
int trl_conv1_direct_bwdw_workspace(int M, int K, int Cout);          // floats
int trl_conv1_direct_bwdw(const ConvSrc& cv, const float* dy, const float* y_gate, int gate_act, float* dw, float* db,
                          float* workspace, int M, int K, int Cout, hipStream_t stream);
// out[e] = sum_s part[s][e] (k_gemm.hip's fixed-order fold; second segment for the bias gradient)
int trl_fold_partials(const float* part, float* out, int n, const float* part2, float* out2, int n2, int splits,
                      hipStream_t stream);
