// K16 -- convolution layers of CNNBase (torchrl/networks/base.py:59-107) as explicit
// im2col + the fp32 MFMA GEMM family of k_gemm.hip.
//
// Activations are kept channels-last, (B, H, W, C) fp32; a conv layer is
//     cols[(b, oy, ox)][c*kh*kw + i*kw + j] = in[b, oy*sh + i, ox*sw + j, c]
//     y[(b, oy, ox)][cout] = act( cols . W[cout][c*kh*kw + i*kw + j]^T + bias )     (trl_linear_fwd_f32)
// so the nn.Conv2d weight (Cout, Cin, kh, kw) is used as stored -- no repacking.  The first layer
// reads the replay buffer's uint8 NCHW frame stacks directly and applies ScaledFloatFrame's
// x/255 - 0.5 on the fly (torchrl/env/atari_wrapper.py:230-240), so frames stay 1 byte in HBM.
// Backward: dW / db through trl_linear_bwd_weight_f32 on the kept cols, d(input) through
// trl_linear_bwd_input_f32 followed by a gather-form col2im (deterministic, no atomics).
// The (B, P, C) <-> (B, C, P) transposes reproduce PyTorch's NCHW flatten order in front of the
// first fully connected layer.  All four kernels are HBM-bound streaming passes.
#include "trl_common.h"

#define CV_THREADS 256

struct ConvGeom { int B, C, H, W, kh, kw, sh, sw, Ho, Wo; };

template <bool U8_NCHW>
__global__ __launch_bounds__(CV_THREADS) void im2col_kernel(const void* __restrict__ in_, float* __restrict__ cols,
                                                            ConvGeom g, float scale, float shift) {
  const int K = g.C * g.kh * g.kw;
  const int64_t total = (int64_t)g.B * g.Ho * g.Wo * K;
  for (int64_t e = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * CV_THREADS) {
    const int k = (int)(e % K);
    const int64_t row = e / K;
    const int ox = (int)(row % g.Wo), oy = (int)((row / g.Wo) % g.Ho), b = (int)(row / ((int64_t)g.Wo * g.Ho));
    const int j = k % g.kw, i = (k / g.kw) % g.kh, c = k / (g.kw * g.kh);
    const int y = oy * g.sh + i, x = ox * g.sw + j;
    float v;
    if (U8_NCHW) v = (float)((const uint8_t*)in_)[(((int64_t)b * g.C + c) * g.H + y) * g.W + x] * scale + shift;
    else         v = ((const float*)in_)[(((int64_t)b * g.H + y) * g.W + x) * g.C + c];
    cols[e] = v;
  }
}

// dx[b, y, x, c] = sum over the windows that cover (y, x)
__global__ __launch_bounds__(CV_THREADS) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx,
                                                            ConvGeom g) {
  const int K = g.C * g.kh * g.kw;
  const int64_t total = (int64_t)g.B * g.H * g.W * g.C;
  for (int64_t e = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * CV_THREADS) {
    const int c = (int)(e % g.C);
    const int64_t pix = e / g.C;
    const int x = (int)(pix % g.W), y = (int)((pix / g.W) % g.H), b = (int)(pix / ((int64_t)g.W * g.H));
    float acc = 0.0f;
    for (int i = y % g.sh; i < g.kh; i += g.sh) {
      const int oy = (y - i) / g.sh;
      if (y - i < 0 || oy >= g.Ho) continue;
      for (int j = x % g.sw; j < g.kw; j += g.sw) {
        const int ox = (x - j) / g.sw;
        if (x - j < 0 || ox >= g.Wo) continue;
        acc += dcols[(((int64_t)b * g.Ho + oy) * g.Wo + ox) * K + (c * g.kh + i) * g.kw + j];
      }
    }
    dx[e] = acc;
  }
}

// out[b][c][p] = in[b][p][c]
// (yg, act: optional gate -- out[e] *= act'(yg[e]) with yg laid out like `out`: the backward pass turns d(features) into
// the last conv layer's dZ in the same pass)
__global__ __launch_bounds__(CV_THREADS) void transpose_bpc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                   int B, int P, int C, const float* __restrict__ yg, int act,
                                                                   int gate_like_in) {
  const int64_t total = (int64_t)B * P * C;
  for (int64_t e = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * CV_THREADS) {
    const int p = (int)(e % P), c = (int)((e / P) % C), b = (int)(e / ((int64_t)P * C));
    const int64_t src = ((int64_t)b * P + p) * C + c;
    float v = in[src];
    if (yg) { const float y = yg[gate_like_in ? src : e]; v *= act == TRL_ACT_TANH ? 1.0f - y * y : (act == TRL_ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : 1.0f); }
    out[e] = v;
  }
}

static int check_geom(const char* who, int B, int C, int H, int W, int kh, int kw, int sh, int sw) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || kh > H || kw > W) {
    trl_set_error("%s: bad geometry", who);
    return TRL_EINVAL;
  }
  return TRL_OK;
}
static int grid_for(int64_t total) {
  int64_t g = (total + CV_THREADS - 1) / CV_THREADS;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// The same through LDS, one (P x C) matrix per workgroup pass: reads of `in` (and of a gate laid out like it) and
// writes of `out` are both contiguous; the naive kernel's reads are C floats apart.  Rows padded to an odd stride.
#define TR_LDS_MAX 8192
__global__ __launch_bounds__(CV_THREADS) void transpose_bpc_lds_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                       int B, int P, int C, const float* __restrict__ yg,
                                                                       int act, int gate_like_in) {
  extern __shared__ float sm[];
  const int PC = P * C, ld = C | 1;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* src = in + (size_t)b * PC;
    const float* gs = yg ? yg + (size_t)b * PC : nullptr;
    for (int e = threadIdx.x; e < PC; e += CV_THREADS) {
      float v = src[e];
      if (gs && gate_like_in) {
        const float y = gs[e];
        v *= act == TRL_ACT_TANH ? 1.0f - y * y : (act == TRL_ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : 1.0f);
      }
      const int p = e / C, c = e - p * C;
      sm[p * ld + c] = v;
    }
    __syncthreads();
    float* dst = out + (size_t)b * PC;
    for (int e = threadIdx.x; e < PC; e += CV_THREADS) {
      const int c = e / P, p = e - c * P;
      float v = sm[p * ld + c];
      if (gs && !gate_like_in) {
        const float y = gs[e];
        v *= act == TRL_ACT_TANH ? 1.0f - y * y : (act == TRL_ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : 1.0f);
      }
      dst[e] = v;
    }
    __syncthreads();
  }
}
static int launch_transpose(const float* in, float* out, int B, int P, int C, const float* yg, int act, int gate_like_in,
                            hipStream_t s) {
  const int floats = P * (C | 1);
  if ((int64_t)P * C <= TR_LDS_MAX) {
    hipLaunchKernelGGL(transpose_bpc_lds_kernel, dim3(std::min(B, 4096)), dim3(CV_THREADS), floats * sizeof(float), s, in, out,
                       B, P, C, yg, act, gate_like_in);
  } else {
    hipLaunchKernelGGL(transpose_bpc_kernel, dim3(grid_for((int64_t)B * P * C)), dim3(CV_THREADS), 0, s, in, out, B, P, C, yg,
                       act, gate_like_in);
  }
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}


extern "C" int trl_im2col_f32(const float* in_nhwc, float* cols, int B, int C, int H, int W, int kh, int kw, int sh,
                              int sw, void* stream) {
  int rc = check_geom("im2col", B, C, H, W, kh, kw, sh, sw);
  if (rc) return rc;
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(in_nhwc && cols, "null pointer");
  ConvGeom g{B, C, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1};
  hipLaunchKernelGGL(im2col_kernel<false>, dim3(grid_for((int64_t)B * g.Ho * g.Wo * C * kh * kw)), dim3(CV_THREADS), 0,
                     (hipStream_t)stream, (const void*)in_nhwc, cols, g, 1.0f, 0.0f);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_im2col_u8_nchw(const uint8_t* in_nchw, float* cols, int B, int C, int H, int W, int kh, int kw,
                                  int sh, int sw, float scale, float shift, void* stream) {
  int rc = check_geom("im2col_u8", B, C, H, W, kh, kw, sh, sw);
  if (rc) return rc;
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(in_nchw && cols, "null pointer");
  ConvGeom g{B, C, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1};
  hipLaunchKernelGGL(im2col_kernel<true>, dim3(grid_for((int64_t)B * g.Ho * g.Wo * C * kh * kw)), dim3(CV_THREADS), 0,
                     (hipStream_t)stream, (const void*)in_nchw, cols, g, scale, shift);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_col2im_f32(const float* dcols, float* dx_nhwc, int B, int C, int H, int W, int kh, int kw, int sh,
                              int sw, void* stream) {
  int rc = check_geom("col2im", B, C, H, W, kh, kw, sh, sw);
  if (rc) return rc;
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(dcols && dx_nhwc, "null pointer");
  ConvGeom g{B, C, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1};
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(CV_THREADS), 0, (hipStream_t)stream,
                     dcols, dx_nhwc, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_transpose_bpc_f32(const float* in, float* out, int B, int P, int C, void* stream) {
  TRL_REQUIRE(B >= 0 && P > 0 && C > 0, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(in && out, "null pointer");
  return launch_transpose(in, out, B, P, C, nullptr, TRL_ACT_NONE, 0, (hipStream_t)stream);
}
extern "C" int trl_transpose_bpc_gate_f32(const float* in, const float* y_gate, int gate_act, int gate_like_in, float* out,
                                          int B, int P, int C, void* stream) {
  TRL_REQUIRE(B >= 0 && P > 0 && C > 0, "bad sizes");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(in && out && y_gate, "null pointer");
  TRL_REQUIRE(gate_act == TRL_ACT_TANH || gate_act == TRL_ACT_RELU || gate_act == TRL_ACT_NONE, "unknown activation");
  return launch_transpose(in, out, B, P, C, y_gate, gate_act, gate_like_in, (hipStream_t)stream);
}
